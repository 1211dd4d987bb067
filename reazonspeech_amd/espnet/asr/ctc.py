"""CTC-based helpers of `reazonspeech.espnet.asr` (pkg/espnet-asr/src/ctc.py): `find_blank` (the longest non-speech stretch
of a window, where long audio is cut, :29-58), `get_timings` (:60-75), `find_end_of_segment` (:77-86), `split_text` (:88-101).

What changes underneath: `ctc_decode` (:12-27: `model.asr_model.encode` + `model.asr_model.ctc.softmax`) is one pass of the
HIP front-end + encoder + CTC head (rs_encoder_set_ctc_out); `ctc_segmentation` is this package's restatement
(ctc_segmentation.py).  Everything else follows the reference line by line, including what it feeds the aligner
(softmax probabilities) and the frame -> sample mapping `idx / (frames + 1) * nsamples`."""
import collections

from . import ctc_segmentation

TOKEN_EOS = {'。', '?', '!'}
TOKEN_COMMA = {'、', ','}
TOKEN_PUNC = TOKEN_EOS | TOKEN_COMMA
PHONEMIC_BREAK = 8000
CHARS_PER_SEGMENT = 15


def ctc_decode(model, samples):
    """character probabilities per encoder frame, float32 [T'][vocab] (ctc.py:12-27; no padding is added here)"""
    return model.ctc_posteriors(samples)


def find_blank(model, samples, threshold=0.98):
    """Find the longest no-speech segment of an audio stream (ctc.py:29-58; arXiv:2002.00551)."""
    Blank = collections.namedtuple('Blank', ['start', 'end'])
    blank_id = model.asr_model.blank_id
    nsamples = len(samples)
    lpz = ctc_decode(model, samples)
    blanks = [Blank(nsamples, nsamples)]
    start = None
    nframes = lpz.shape[0]
    for idx, prob in enumerate(lpz.T[blank_id]):
        if prob > threshold:
            if start is None:
                start = int(idx / (nframes + 1) * nsamples)
        else:
            if start and start > 0:
                end = int(idx / (nframes + 1) * nsamples)
                blanks.append(Blank(start, end))
            start = None
    return max(blanks, key=lambda b: b.end - b.start)


def get_timings(model, samples, text):
    """playback time (in samples) of each character by CTC segmentation (ctc.py:60-75)"""
    lpz = ctc_decode(model, samples)
    opt = ctc_segmentation.CtcSegmentationParameters(
        index_duration=len(samples) / (lpz.shape[0] + 1),
        char_list=model.asr_model.token_list[:-1]
    )
    matrix, indices = ctc_segmentation.prepare_text(opt, [text])
    timings = ctc_segmentation.ctc_segmentation(opt, lpz, matrix)[0]
    # "+1" to skip a preceding blank character.
    return timings[indices[0] + 1:indices[1]]


def find_end_of_segment(text, timings, start):
    nchar = len(text)
    idx = start
    for idx in range(start, nchar):
        if idx < nchar - 1:
            cur = text[idx]
            nex = text[idx + 1]
            if nex not in TOKEN_PUNC:
                if cur in TOKEN_EOS:
                    break
                elif idx - start >= CHARS_PER_SEGMENT:
                    if cur in TOKEN_COMMA or timings[idx + 1] - timings[idx] > PHONEMIC_BREAK:
                        break
    return idx


def split_text(model, samples, text):
    """Split a text into (start sample, end sample, text) segments (ctc.py:88-101); one segment spanning the whole window when
    the alignment fails, like the reference's blanket `except Exception`."""
    try:
        timings = get_timings(model, samples, text)
    except Exception:
        return [(0, len(samples), text)]
    ret = []
    start = 0
    while start < len(text):
        end = find_end_of_segment(text, timings, start)
        ret.append((timings[start], timings[end], text[start:end + 1]))
        start = end + 1
    return ret
