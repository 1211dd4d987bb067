"""CTC-based helpers of `reazonspeech.espnet.asr`: where to cut long audio and how to time-stamp a recognised text.

Behaviour contract = the reference's pkg/espnet-asr/src/ctc.py (pinned by tests/golden/reference_espnet.json, which that
file produced on a fake model): `find_blank` (:29-58), `get_timings` (:60-75), `find_end_of_segment` (:77-86), `split_text`
(:88-101), the constants (:6-10).  The implementation is this package's own: the blank stretches come out of a run-length
pass over a numpy mask instead of a per-frame Python loop, the segment end is a forward scan.  Underneath, `ctc_decode`
(:12-27: `model.asr_model.encode` + `model.asr_model.ctc.softmax`) is one pass of the HIP front-end + encoder + CTC head
(rs_encoder_set_ctc_out) and `ctc_segmentation` is this package's restatement of the third-party aligner
(ctc_segmentation.py)."""
import collections

import numpy as np

from . import ctc_segmentation

TOKEN_EOS = {'。', '?', '!'}
TOKEN_COMMA = {'、', ','}
TOKEN_PUNC = TOKEN_EOS | TOKEN_COMMA
PHONEMIC_BREAK = 8000
CHARS_PER_SEGMENT = 15

Blank = collections.namedtuple('Blank', ['start', 'end'])


def ctc_decode(model, samples):
    """character probabilities per encoder frame, float32 [T'][vocab] (ctc.py:12-27; no padding is added here)"""
    return model.ctc_posteriors(samples)


def _frame_to_sample(frame, n_frames, n_samples):
    """the reference's frame -> sample map (ctc.py:50,53): int(idx / (frames + 1) * nsamples), in float64 like Python's"""
    return (np.asarray(frame, np.float64) / (n_frames + 1) * n_samples).astype(np.int64)


def find_blank(model, samples, threshold=0.98):
    """The longest stretch of the window in which the CTC head is sure of silence (blank posterior > threshold): long audio
    is cut in its middle (arXiv:2002.00551).  -> Blank(start, end) in samples.

    What counts as a stretch follows the reference's scan exactly: it must be CLOSED by a frame below the threshold (one
    that runs to the end of the window is ignored), its first sample must be > 0 (one that starts at frame 0 is ignored), and
    when nothing qualifies — or on a tie in length — the earliest candidate wins, the first being the empty stretch at the
    very end of the window (nsamples, nsamples)."""
    n_samples = len(samples)
    post = ctc_decode(model, samples)
    n_frames = post.shape[0]
    silent = np.asarray(post[:, model.asr_model.blank_id] > threshold)
    # run-length pass: +1 where a silent run begins, -1 at the first frame after it
    edges = np.diff(np.concatenate(([0], silent.astype(np.int8), [0])))
    first = np.flatnonzero(edges == 1)
    after = np.flatnonzero(edges == -1)
    closed = after < n_frames
    begin = _frame_to_sample(first[closed], n_frames, n_samples)
    end = _frame_to_sample(after[closed], n_frames, n_samples)
    usable = begin > 0
    begin, end = begin[usable], end[usable]
    if begin.size:
        k = int(np.argmax(end - begin))            # first of the longest
        if end[k] - begin[k] > 0:
            return Blank(int(begin[k]), int(end[k]))
    return Blank(n_samples, n_samples)


def get_timings(model, samples, text):
    """sample position of every character of `text`, by CTC segmentation of the window's posteriors (ctc.py:60-75): one
    aligner index lasts len(samples) / (frames + 1) samples, the token list goes in without its last entry (<sos/eos>), and
    the utterance's own leading separator is skipped (the '+ 1')."""
    post = ctc_decode(model, samples)
    params = ctc_segmentation.CtcSegmentationParameters(index_duration=len(samples) / (post.shape[0] + 1),
                                                        char_list=model.asr_model.token_list[:-1])
    ground_truth, bounds = ctc_segmentation.prepare_text(params, [text])
    per_symbol = ctc_segmentation.ctc_segmentation(params, post, ground_truth)[0]
    return per_symbol[bounds[0] + 1:bounds[1]]


def find_end_of_segment(text, timings, start):
    """index of the last character of the segment that begins at `start` (ctc.py:77-86).  A segment closes after a sentence
    end (。?!); once it holds CHARS_PER_SEGMENT characters it also closes after a comma or in front of a pause longer than
    PHONEMIC_BREAK samples; it never closes in front of punctuation; the text's last character closes whatever is open."""
    last = len(text) - 1
    pos = start
    while pos < last:
        if text[pos + 1] not in TOKEN_PUNC:
            here = text[pos]
            if here in TOKEN_EOS:
                break
            long_enough = pos - start >= CHARS_PER_SEGMENT
            if long_enough and (here in TOKEN_COMMA or timings[pos + 1] - timings[pos] > PHONEMIC_BREAK):
                break
        pos += 1
    return pos


def split_text(model, samples, text):
    """[(start sample, end sample, text)] segments of a recognised window (ctc.py:88-101).  When the alignment cannot be
    computed — the reference catches every exception there — the whole window is one segment."""
    try:
        timings = get_timings(model, samples, text)
    except Exception:
        return [(0, len(samples), text)]
    pieces = []
    head = 0
    while head < len(text):
        tail = find_end_of_segment(text, timings, head)
        pieces.append((timings[head], timings[tail], text[head:tail + 1]))
        head = tail + 1
    return pieces
