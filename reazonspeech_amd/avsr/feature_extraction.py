"""`AVHubertFeatureExtractor` of `reazonspeech.avsr` (pkg/avsr/src/avhubert/feature_extraction_avhubert.py:16-241): raw audio and
grey mouth-region video -> `input_values` [B][T][104], `pixel_values` [B][T][1][88][88], `padding_mask` [B][T].  Host code (numpy);
the model runs on the device.

What is restated here: log filterbank features + stacking + per-frame layer norm (:120-137, :226), the image transform (centre crop,
/ 255, normalise: :37-44), the audio / video time alignment (:139-158) and batch padding (:199-232).  What needs packages this image
does not have raises a clear error instead of guessing: decoding files (cv2 / librosa) and mouth extraction (mediapipe, :77-118) —
pass arrays of already extracted grey mouth crops and 16 kHz samples.

`from_pretrained(dir)` of both classes reads what the reference's `save_pretrained` writes: `preprocessor_config.json` (the
extractor's constructor arguments) and the tokenizer files of a `PreTrainedTokenizerFast` (processing_avhubert.py:24-25)."""
import json
import os

import numpy as np

IMAGE_MEAN, IMAGE_STD = 0.421, 0.165


def _hz2mel(hz):
    return 2595.0 * np.log10(1.0 + hz / 700.0)


def _mel2hz(mel):
    return 700.0 * (10.0 ** (mel / 2595.0) - 1.0)


def logfbank(signal, samplerate=16000, winlen=0.025, winstep=0.01, nfilt=26, nfft=512, preemph=0.97):
    """[UPSTREAM] python_speech_features.logfbank with its defaults (the reference calls `logfbank(audio, samplerate=sr)`, :135):
    pre-emphasis 0.97 over the whole signal, 25 ms frames every 10 ms (the tail zero-padded to a whole frame, rectangular window),
    |rfft(512)|^2 / 512, 26 triangular filters spaced on the 2595 log10(1 + f / 700) mel scale between 0 and sr / 2 with FFT-bin
    edges floor((nfft + 1) f / sr), zero energies replaced by machine epsilon, natural log.  -> float64 [frames][26]"""
    signal = np.asarray(signal, dtype=np.float64)
    signal = np.append(signal[0], signal[1:] - preemph * signal[:-1])
    flen, fstep = int(round(winlen * samplerate)), int(round(winstep * samplerate))
    n = len(signal)
    frames = 1 if n <= flen else 1 + int(np.ceil((n - flen) / fstep))
    padded = np.concatenate([signal, np.zeros(((frames - 1) * fstep + flen - n,))])
    idx = np.arange(flen)[None, :] + (np.arange(frames) * fstep)[:, None]
    pspec = np.square(np.abs(np.fft.rfft(padded[idx], nfft))) / nfft
    melpoints = np.linspace(_hz2mel(0.0), _hz2mel(samplerate / 2.0), nfilt + 2)
    bins = np.floor((nfft + 1) * _mel2hz(melpoints) / samplerate)
    fb = np.zeros((nfilt, nfft // 2 + 1))
    for j in range(nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fb[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fb[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    feat = pspec @ fb.T
    return np.log(np.where(feat == 0, np.finfo(float).eps, feat))


class AVHubertFeatureExtractor:
    model_input_names = ["input_values", "pixel_values"]

    def __init__(self, max_sample_size=None, normalize=True, stack_order_audio=4, image_crop_size=88, image_mean=IMAGE_MEAN, image_std=IMAGE_STD,
                 sr=16000, **kwargs):
        self.max_sample_size, self.normalize, self.stack_order_audio = max_sample_size, normalize, stack_order_audio
        self.image_crop_size, self.image_mean, self.image_std, self.sr = image_crop_size, image_mean, image_std, sr

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        """a directory (or the file itself) with `preprocessor_config.json` as FeatureExtractionMixin.save_pretrained writes it:
        the constructor's arguments plus bookkeeping keys (`feature_extractor_type`, `processor_class`, the printed `transforms`)"""
        f = path if os.path.isfile(path) else os.path.join(path, "preprocessor_config.json")
        if not os.path.isfile(f):
            raise FileNotFoundError(f"{f}: no preprocessor_config.json (the reference loads it with FeatureExtractionMixin.from_pretrained)")
        with open(f, encoding="utf-8") as fp:
            raw = json.load(fp)
        known = ("max_sample_size", "normalize", "stack_order_audio", "image_crop_size", "image_mean", "image_std", "sr")
        args = {k: raw[k] for k in known if k in raw}
        args.update(kwargs)
        return cls(**args)

    def _load_audio(self, audio):
        if isinstance(audio, str):
            raise RuntimeError("decoding an audio file needs librosa (feature_extraction_avhubert.py:131), which this image does not have: "
                               "pass the 16 kHz samples as a float array")
        fb = logfbank(np.asarray(audio, dtype=np.float32), samplerate=self.sr).astype(np.float32)
        k = self.stack_order_audio                                        # :121-129 stacker
        if len(fb) % k:
            fb = np.concatenate([fb, np.zeros((k - len(fb) % k, fb.shape[1]), fb.dtype)], axis=0)
        return fb.reshape(-1, k * fb.shape[1])

    def _load_video(self, video, extract_mouth=False):
        if isinstance(video, str) or extract_mouth:
            raise RuntimeError("decoding a video file / extracting the mouth region needs cv2 and mediapipe "
                               "(feature_extraction_avhubert.py:55-118), which this image does not have: pass grey mouth crops uint8 [T][H][W]")
        v = np.asarray(video)
        if v.ndim == 4:
            # colour frames -> grey (:72-73 cv2.cvtColor(frame, COLOR_BGR2GRAY)): [UPSTREAM] OpenCV's 8-bit path is fixed point,
            # (1868 B + 9617 G + 4899 R + 2^13) >> 14, not a float product — the two differ by one grey level on some pixels
            c = v.astype(np.int64)
            v = ((1868 * c[..., 0] + 9617 * c[..., 1] + 4899 * c[..., 2] + 8192) >> 14).astype(np.uint8)
        return v[:, None]                                                 # [T][1][H][W]

    def _transform(self, frames):
        """:37-44: CenterCrop(image_crop_size), scale to [0, 1], Normalize(mean, std)"""
        c = self.image_crop_size
        h, w = frames.shape[-2:]
        top, left = int(round((h - c) / 2.0)), int(round((w - c) / 2.0))
        x = frames[..., top:top + c, left:left + c].astype(np.float32)
        if frames.dtype == np.uint8:
            x = x / 255.0
        return (x - self.image_mean) / self.image_std

    def __call__(self, raw_audio=None, raw_video=None, extract_mouth=False, **kwargs):
        raw_audio = raw_audio if isinstance(raw_audio, list) else [raw_audio]
        raw_video = raw_video if isinstance(raw_video, list) else [raw_video]
        audio = [self._load_audio(a) if a is not None else None for a in raw_audio]
        video = [self._load_video(v, extract_mouth) if v is not None else None for v in raw_video]
        c = self.image_crop_size
        for i in range(len(audio)):
            assert audio[i] is not None or video[i] is not None
            if audio[i] is None:
                audio[i] = np.zeros((video[i].shape[0], 26 * self.stack_order_audio), np.float32)
            elif video[i] is None:
                video[i] = np.zeros((audio[i].shape[0], 1, c, c), np.uint8)
        for i, (a, v) in enumerate(zip(audio, video)):                    # :139-158 nearest-frame alignment of the video to the audio rate
            if len(a) != len(v):
                idx = np.minimum(np.floor(np.arange(len(a), dtype=np.float32) * len(v) / len(a)), len(v) - 1).astype(np.int64)
                video[i] = v[idx]
        T = max(len(a) for a in audio)
        iv, pv, pm = [], [], []
        for a, v in zip(audio, video):
            rem = T - len(a)
            a = np.concatenate([a, np.zeros((rem,) + a.shape[1:], a.dtype)])
            v = np.concatenate([v, np.zeros((rem,) + v.shape[1:], v.dtype)])
            if self.max_sample_size:
                a, v = a[:self.max_sample_size], v[:self.max_sample_size]
            m = np.zeros((T,), np.float32)
            m[T - rem:] = 1.0
            iv.append(a); pv.append(v); pm.append(m)
        iv = np.stack(iv).astype(np.float32)
        if self.normalize:                                                # :226 F.layer_norm over the feature axis, no affine
            mu = iv.mean(-1, keepdims=True)
            iv = (iv - mu) / np.sqrt(((iv - mu) ** 2).mean(-1, keepdims=True) + 1e-5)
        return {"input_values": iv, "pixel_values": self._transform(np.stack(pv)).astype(np.float32), "padding_mask": np.stack(pm)}


class _FastTokenizer:
    """the few calls the processor makes on its `PreTrainedTokenizerFast`, over the `tokenizers` library alone (used when
    transformers cannot be imported): decode / batch_decode with `skip_special_tokens`, and `__call__` on a list of texts ->
    right-padded `input_ids` / `attention_mask` (numpy int64)"""

    def __init__(self, path):
        from tokenizers import Tokenizer
        self.tk = Tokenizer.from_file(os.path.join(path, "tokenizer.json"))
        self.pad_id = 0
        for name in ("tokenizer_config.json", "special_tokens_map.json"):
            f = os.path.join(path, name)
            if os.path.isfile(f):
                with open(f, encoding="utf-8") as fp:
                    pad = json.load(fp).get("pad_token")
                pad = pad.get("content") if isinstance(pad, dict) else pad
                if pad is not None and self.tk.token_to_id(pad) is not None:
                    self.pad_id = self.tk.token_to_id(pad)
                    break

    def decode(self, ids, skip_special_tokens=False, **kwargs):
        ids = [int(i) for i in (ids.tolist() if hasattr(ids, "tolist") else ids)]
        return self.tk.decode(ids, skip_special_tokens=skip_special_tokens)

    def batch_decode(self, seqs, skip_special_tokens=False, **kwargs):
        return [self.decode(s, skip_special_tokens=skip_special_tokens) for s in seqs]

    def __call__(self, text, **kwargs):
        enc = [self.tk.encode(t, add_special_tokens=False).ids for t in text]
        L = max(len(e) for e in enc)
        ids = np.full((len(enc), L), self.pad_id, np.int64)
        mask = np.zeros((len(enc), L), np.int64)
        for i, e in enumerate(enc):
            ids[i, :len(e)] = e
            mask[i, :len(e)] = 1
        return {"input_ids": ids, "attention_mask": mask}


def wrap_targets(text):
    """processing_avhubert.py:58-73: every target text is sent to the tokenizer as `<s>` + text + `</s>` (whichever end is missing)"""
    out = []
    for t in text:
        if not t.startswith("<s>"):
            t = "<s>" + t
        if not t.endswith("</s>"):
            t = t + "</s>"
        out.append(t)
    return out


class AVHubertProcessor:
    """processing_avhubert.py:8-118: the feature extractor plus a tokenizer (a `PreTrainedTokenizerFast` in the reference):
    `__call__` makes the model inputs (and, given `text`, the teacher-forcing tensors), `decode` / `batch_decode` are the tokenizer's"""

    def __init__(self, feature_extractor=None, tokenizer=None):
        self.feature_extractor = feature_extractor or AVHubertFeatureExtractor()
        self.tokenizer = tokenizer

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        """README.rst: `AVHubertProcessor.from_pretrained("path/to/avsr/model")` — `preprocessor_config.json` + the tokenizer files"""
        fe = AVHubertFeatureExtractor.from_pretrained(path, **kwargs)
        if not os.path.isfile(os.path.join(path, "tokenizer.json")):
            raise FileNotFoundError(f"{path}: no tokenizer.json (the processor's tokenizer_class is PreTrainedTokenizerFast)")
        try:
            from transformers import PreTrainedTokenizerFast
            tok = PreTrainedTokenizerFast.from_pretrained(path)
        except ImportError:
            tok = _FastTokenizer(path)
        return cls(fe, tok)

    def __call__(self, raw_audio=None, raw_video=None, text=None, **kwargs):
        is_batched = isinstance(raw_audio, list)
        if raw_audio is None and raw_video is None and text is None:
            raise ValueError("You need to specify either an `raw_audio`, `raw_video` or `text` input to process.")
        inputs = None
        if raw_audio is not None or raw_video is not None:
            inputs = self.feature_extractor(raw_audio, raw_video, **kwargs)
        if text is None:
            return inputs
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer was given to this processor")
        if not is_batched:                                                 # (:55-56: decided by `raw_audio`, like the reference)
            text = [text]
        kwargs.pop("extract_mouth", None)
        kwargs.setdefault("return_tensors", "np" if isinstance(self.tokenizer, _FastTokenizer) else "pt")
        enc = self.tokenizer(wrap_targets(text), **kwargs)
        if inputs is None:
            return enc
        ids, mask = enc["input_ids"], enc["attention_mask"]               # :84-87 teacher forcing: inputs drop the last token, labels the first
        inputs["decoder_input_ids"] = ids[:, :-1].copy() if isinstance(ids, np.ndarray) else ids[:, :-1].clone()
        inputs["decoder_attention_mask"] = mask[:, :-1]
        inputs["labels"] = ids[:, 1:]
        return inputs

    def decode(self, *args, **kwargs):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer was given to this processor")
        return self.tokenizer.decode(*args, **kwargs)

    def batch_decode(self, *args, **kwargs):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer was given to this processor")
        return self.tokenizer.batch_decode(*args, **kwargs)
