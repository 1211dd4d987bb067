"""Synthetic model inputs of the AV-HuBERT path, in the form `AVHubertFeatureExtractor.__call__` hands them to the model
(pkg/avsr/src/avhubert/feature_extraction_avhubert.py:199-233): input_values float32 [B][T][104] (four stacked 26-dim log
filterbank frames, layer-normalised per frame), pixel_values float32 [B][T][1][88][88] (grey mouth crops, (x / 255 - 0.421) / 0.165),
padding_mask float32 [B][T] (1 = padding).  Padded frames carry what the extractor puts there: zero audio features and the
normalised value of a black image."""
import numpy as np

IMAGE_MEAN, IMAGE_STD = 0.421, 0.165


def synthetic_clips(batch: int, frames: int, seed: int = 0, ragged: bool = False, min_frames: int = 8, image_size: int = 88, feat_dim: int = 104):
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_frames, frames + 1, size=batch) if ragged else np.full((batch,), frames)
    if ragged:
        lens[int(rng.integers(0, batch))] = frames
    audio = rng.standard_normal((batch, frames, feat_dim)).astype(np.float32)
    audio = (audio - audio.mean(-1, keepdims=True)) / np.sqrt(audio.var(-1, keepdims=True) + 1e-5)
    # "mouth": a bright blob that opens and closes on a textured face, one phase / rate per clip
    yy, xx = np.mgrid[0:image_size, 0:image_size].astype(np.float32)
    video = np.empty((batch, frames, 1, image_size, image_size), np.float32)
    for b in range(batch):
        rate, phase = rng.uniform(0.2, 0.9), rng.uniform(0, 6.28)
        tex = rng.uniform(0.25, 0.6, size=(image_size, image_size)).astype(np.float32)
        cx, cy = image_size / 2 + rng.uniform(-6, 6), image_size / 2 + rng.uniform(-6, 6)
        for t in range(frames):
            opening = 4.0 + 10.0 * (0.5 + 0.5 * np.sin(rate * t + phase))
            blob = np.exp(-(((xx - cx) / 22.0) ** 2 + ((yy - cy) / opening) ** 2))
            img = np.clip(tex * 0.6 + 0.55 * blob + 0.02 * rng.standard_normal((image_size, image_size)), 0.0, 1.0)
            video[b, t, 0] = (np.round(img * 255.0) / 255.0 - IMAGE_MEAN) / IMAGE_STD
    mask = np.zeros((batch, frames), np.float32)
    for b in range(batch):
        n = int(lens[b])
        audio[b, n:] = 0.0
        video[b, n:] = (0.0 - IMAGE_MEAN) / IMAGE_STD
        mask[b, n:] = 1.0
    return audio, video.astype(np.float32), mask, lens.astype(np.int32)
