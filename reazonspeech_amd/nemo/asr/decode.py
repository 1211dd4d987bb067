"""Hypothesis -> text / subword timestamps / segments.

Restates pkg/nemo-asr/src/decode.py:4-66.  It is the one part of the hot path whose
behaviour is fully pinned by in-tree reference code, so `tests/test_host_reference_parity.py`
checks it against golden vectors produced by importing the reference file itself
(`tests/golden/make_reference_golden.py`).
"""
import weakref

from .interface import Subword, Segment, TranscribeResult

# decode.py:4-7
PAD_SECONDS = 0.5
SECONDS_PER_STEP = 0.08
SUBWORDS_PER_SEGMENTS = 10
PHONEMIC_BREAK = 0.5

# decode.py:9-11
TOKEN_EOS = {'。', '?', '!'}
TOKEN_COMMA = {'、', ','}
TOKEN_PUNC = TOKEN_EOS | TOKEN_COMMA


def find_end_of_segment(subwords, start):
    """Index of the last subword of the segment beginning at `start` (decode.py:13-26).

    A segment closes after a sentence-final mark, or — once it holds at least
    SUBWORDS_PER_SEGMENTS subwords — after a comma or before a pause longer than
    PHONEMIC_BREAK; it never closes when the next token is punctuation, and the last
    subword always closes the final segment.
    """
    last = len(subwords) - 1
    idx = start
    while idx < last:
        cur, nxt = subwords[idx], subwords[idx + 1]
        if nxt.token not in TOKEN_PUNC:
            if cur.token in TOKEN_EOS:
                return idx
            if idx - start >= SUBWORDS_PER_SEGMENTS and (
                    cur.token in TOKEN_COMMA or nxt.seconds - cur.seconds > PHONEMIC_BREAK):
                return idx
        idx += 1
    return idx


_piece_text_cache = weakref.WeakKeyDictionary()


def _piece_text(tokenizer):
    """`lambda token_id: tokenizer.ids_to_text([token_id])` with the answers remembered per tokenizer: the reference asks
    the tokenizer once per emitted token (decode.py:49), a vocabulary has ~3000 entries and a batch ~12 000 tokens."""
    try:
        cache = _piece_text_cache.setdefault(tokenizer, {})
    except TypeError:                      # a tokenizer that cannot be weakly referenced: no memory across calls
        cache = {}

    def get(token_id):
        text = cache.get(token_id)
        if text is None:
            text = cache[token_id] = tokenizer.ids_to_text([token_id])
        return text
    return get


def decode_hypothesis(model, hyp):
    """Build a TranscribeResult from an ALSD-shaped hypothesis (decode.py:28-66).

    `model` only has to provide `.tokenizer.ids_to_text(list[int]) -> str`.
    """
    ids = hyp.y_sequence.tolist()[1:]          # decode.py:40 — drop the leading blank
    text = model.tokenizer.ids_to_text(ids)     # decode.py:41

    piece = _piece_text(model.tokenizer)
    subwords = []
    for idx, (token_id, step) in enumerate(zip(ids, hyp.timestamp)):
        token = piece(token_id)
        if not token:
            continue                            # bare U+2581 pieces decode to "" and are dropped AFTER idx was assigned (decode.py:53)
        seconds = max(SECONDS_PER_STEP * (step - idx - 1) - PAD_SECONDS, 0)   # decode.py:48
        subwords.append(Subword(seconds, token_id, token))

    segments = []
    start = 0
    while start < len(subwords):
        end = find_end_of_segment(subwords, start)
        segments.append(Segment(
            start_seconds=subwords[start].seconds,
            end_seconds=subwords[end].seconds + SECONDS_PER_STEP,   # decode.py:61
            text="".join(sw.token for sw in subwords[start:end + 1]),
        ))
        start = end + 1

    return TranscribeResult(text, subwords, segments)
