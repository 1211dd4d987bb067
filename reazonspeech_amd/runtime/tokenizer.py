"""Tokenizers: the reference reads `model.tokenizer.ids_to_text` of NeMo's SentencePiece
wrapper (pkg/nemo-asr/src/decode.py:41,47).  Two implementations with that one method:

  SentencePieceTokenizer   wraps a `tokenizer.model` taken from a .nemo archive
  SyntheticTokenizer       a seeded vocabulary for synthetic-weight runs (no checkpoint here)
"""
import numpy as np

_WS = "▁"


class SentencePieceTokenizer:
    def __init__(self, model_bytes: bytes):
        import sentencepiece as spm
        self._sp = spm.SentencePieceProcessor(model_proto=model_bytes)
        self.vocab_size = self._sp.get_piece_size()

    def ids_to_text(self, ids):
        return self._sp.decode([int(i) for i in ids])

    def ids_to_tokens(self, ids):
        return [self._sp.id_to_piece(int(i)) for i in ids]


class SyntheticTokenizer:
    """`vocab_size` pieces drawn from kana / kanji / punctuation, with SentencePiece's decode
    rule: concatenate pieces, U+2581 -> space, strip the leading space.  Piece 0 is a bare
    U+2581 (decodes to "" on its own and is dropped by decode.py:53), pieces 1..5 are the
    punctuation marks the segmenter looks for (decode.py:9-11)."""

    def __init__(self, vocab_size: int, seed: int = 0):
        rng = np.random.default_rng(seed)
        fixed = [_WS, "。", "、", "?", "!", ","]
        pool = [chr(c) for c in range(0x3041, 0x3097)] + [chr(c) for c in range(0x30A1, 0x30FB)] + \
               [chr(c) for c in range(0x4E00, 0x4E00 + 4096)]
        pieces = list(fixed[:vocab_size])
        seen = set(pieces)
        while len(pieces) < vocab_size:
            n = int(rng.integers(1, 3))
            p = "".join(pool[int(i)] for i in rng.integers(0, len(pool), size=n))
            if rng.random() < 0.1:
                p = _WS + p
            if p not in seen:
                seen.add(p)
                pieces.append(p)
        self.pieces = pieces
        self.vocab_size = vocab_size

    def ids_to_tokens(self, ids):
        return [self.pieces[int(i)] for i in ids]

    def ids_to_text(self, ids):
        text = "".join(self.pieces[int(i)] for i in ids).replace(_WS, " ")
        return text[1:] if text.startswith(" ") else text
