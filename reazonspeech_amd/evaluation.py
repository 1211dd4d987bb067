"""CER evaluation with the batched MI355X path — SURVEY.md §8f "next" row 1.

The reference harness (`reazonspeech.evaluation`, pkg/evaluation/src/base.py:144-230) maps
`_evaluate` / `_evaluate_batch` over an HF dataset and spreads work over GPUs with
`datasets.map(with_rank=True)` + `cuda:{rank % num_gpus}` (base.py:194-212,
examples/rs-nemo/eval.py:24-28); its nemo example never implemented the batched hook
(`_evaluate_batch` raises, eval.py:31-32).  Here the batched hook is the natural caller of
`transcribe_batch`, and multi-GPU is one process per GPU through `torch.distributed`
(`runtime/dist.py`): every rank transcribes a length-sorted shard, rank 0 gathers predictions.

`normalize` / `calculate_cer` restate pkg/evaluation/src/utils.py:16-33 (punctuation stripped,
full-width alphanumerics folded, digits spelled out with num2words when that package exists;
`editdistance` is replaced by an in-file Levenshtein).  Pinned by tests/golden/reference_eval.json.
"""
import json
import re
from typing import Any, Dict, Iterable, List, Optional, TypedDict

from .nemo.asr import TranscribeConfig, audio_from_numpy, audio_from_path, load_model, transcribe, transcribe_batch
from .runtime import dist as rdist


class CERResult(TypedDict):
    cer: float
    distance: int
    length: int


class EvaluationResult(TypedDict):
    prediction: str


class EvaluationResultBatch(TypedDict):
    predictions: List[str]


# utils.py:16-19
PUNCTUATIONS = {ord(x): "" for x in "、。「」『』，,？！!!?!?"}
ZENKAKU = "ａｂｃｄｅｆｇｈｉｊｋｌｍｎｏｐｑｒｓｔｕｖｗｘｙｚＡＢＣＤＥＦＧＨＩＪＫＬＭＮＯＰＱＲＳＴＵＶＷＸＹＺ０１２３４５６７８９"
HANKAKU = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
ZEN2HAN = str.maketrans(ZENKAKU, HANKAKU)


def normalize(s: str) -> str:
    """utils.py:22-28.  Without `num2words` installed digits are left as they are."""
    s = s.translate(PUNCTUATIONS).translate(ZEN2HAN)
    try:
        import num2words
    except ImportError:
        return s
    try:
        return re.sub(r"\d+\.?\d*", lambda m: num2words.num2words(m.group(0), lang="ja"), s)
    except OverflowError:
        return s


def edit_distance(a: str, b: str) -> int:
    """Levenshtein distance (what `editdistance.eval` computes, utils.py:32)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def calculate_cer(reference: str, prediction: str) -> CERResult:
    """utils.py:29-33"""
    reference = normalize(reference)
    prediction = normalize(prediction)
    distance = edit_distance(reference, prediction)
    return CERResult(cer=distance / len(reference), distance=distance, length=len(reference))


def _audio_of(example: Dict[str, Any]):
    audio = example["audio"]
    if isinstance(audio, dict):
        if audio.get("array") is not None:
            return audio_from_numpy(audio["array"], audio["sampling_rate"])
        return audio_from_path(audio["path"])
    return audio_from_path(audio)


class RSAmdEvaluator:
    """Counterpart of `RSNemoEvaluator` (examples/rs-nemo/eval.py:15-32) with a working batch hook."""

    def __init__(self, model=None, dataset=None, output_file=None, batch_size: Optional[int] = 256,
                 text_column: str = "text"):
        self.model = model
        self.dataset = dataset
        self.output_file = output_file
        self.batch_size = batch_size
        self.text_column = text_column
        self.config = TranscribeConfig(verbose=False)

    def _ensure_model(self, rank: Optional[int], num_gpus: Optional[int]):
        if self.model is None:
            rank = 0 if rank is None else rank
            num_gpus = 1 if num_gpus is None else num_gpus
            self.model = self._load_model(device=f"cuda:{rank % num_gpus}")     # eval.py:24-27

    def _evaluate(self, example, rank: Optional[int] = None, num_gpus: Optional[int] = None, **kwargs) -> EvaluationResult:
        self._ensure_model(rank, num_gpus)
        return {"prediction": self._transcribe(self.model, _audio_of(example), self.config).text}

    def _evaluate_batch(self, batch: Dict[str, List[Any]], rank: Optional[int] = None, num_gpus: Optional[int] = None,
                        **kwargs) -> EvaluationResultBatch:
        """`batch` is column-major like `datasets.map(batched=True)` hands it over (base.py:205-212)."""
        self._ensure_model(rank, num_gpus)
        audios = [_audio_of({"audio": a}) for a in batch["audio"]]
        return {"predictions": [r.text for r in self._transcribe_batch(self.model, audios, self.config)]}

    # the three package functions the hooks go through (the espnet evaluator below swaps them)
    _load_model = staticmethod(load_model)
    _transcribe = staticmethod(transcribe)
    _transcribe_batch = staticmethod(transcribe_batch)

    def evaluate(self, dataset: Optional[Iterable[Dict[str, Any]]] = None, batch_size: Optional[int] = None,
                 text_column: Optional[str] = None, output_file=None) -> List[Dict[str, Any]]:
        """Transcribe every example, attach prediction / distance / length, print the corpus CER
        (base.py:144-230).  With `torch.distributed` initialised each rank handles a length-balanced
        shard and rank 0 returns the merged rows (other ranks return their own shard)."""
        rows = list(dataset if dataset is not None else self.dataset or [])
        if not rows and dataset is None and self.dataset is None:
            raise ValueError("No dataset provided and self.dataset is None.")
        batch_size = batch_size or self.batch_size
        text_column = text_column or self.text_column
        world, rank = rdist.world_size(), rdist.rank()
        mine = list(range(len(rows)))[rank::world] if world > 1 else list(range(len(rows)))
        preds: Dict[int, str] = {}
        if batch_size is None:
            for i in mine:
                preds[i] = self._evaluate(rows[i], rank=rank, num_gpus=world)["prediction"]
        else:
            for lo in range(0, len(mine), batch_size):
                idx = mine[lo:lo + batch_size]
                out = self._evaluate_batch({"audio": [rows[i]["audio"] for i in idx]}, rank=rank, num_gpus=world)
                preds.update(zip(idx, out["predictions"]))
        if world > 1:
            import torch.distributed as dist
            gathered: List[Optional[Dict[int, str]]] = [None] * world
            dist.all_gather_object(gathered, preds)
            if rank == 0:
                for g in gathered:
                    preds.update(g)
        evaluated = []
        for i in sorted(preds):
            row = dict(rows[i])
            row["prediction"] = preds[i]
            cer = calculate_cer(row[text_column], row["prediction"])
            row["distance"], row["length"] = cer["distance"], cer["length"]
            evaluated.append(row)
        dist_sum = sum(r["distance"] for r in evaluated)
        length = sum(r["length"] for r in evaluated)
        if rank == 0 and length:
            print(f"CER: {dist_sum / length * 100:.2f}%")
        out_path = output_file or self.output_file
        if out_path is not None and rank == 0:
            with open(out_path, "w", encoding="utf-8") as fp:
                for r in evaluated:
                    slim = {k: v for k, v in r.items() if k != "audio"}
                    fp.write(json.dumps(slim, ensure_ascii=False) + "\n")
        return evaluated


class RSEspnetAmdEvaluator(RSAmdEvaluator):
    """Counterpart of `RSESPNETEvaluator` (pkg/evaluation/examples/rs-espnet/eval.py:16-33) over `reazonspeech.espnet.asr` of this
    package: the same two hooks, with a working batch hook (utterances of at most one 20 s window are recognised as one device
    batch, longer ones go through the windowing loop one by one)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from .espnet.asr import TranscribeConfig as EspnetConfig
        self.config = EspnetConfig(verbose=False)

    @staticmethod
    def _load_model(device=None):
        from .espnet.asr import load_model as lm
        return lm(device=device)

    @staticmethod
    def _transcribe(model, audio, config=None):
        from .espnet.asr import transcribe as tr
        return tr(model, audio, config)

    @staticmethod
    def _transcribe_batch(model, audios, config=None):
        from .espnet.asr import transcribe_batch as tb
        return tb(model, audios, config)


class RSK2AmdEvaluator(RSAmdEvaluator):
    """Counterpart of `RSK2Evaluator` (pkg/evaluation/examples/rs-k2/eval.py:15-33) over `reazonspeech.k2.asr` of this package:
    `_evaluate` is the reference's hook (load on cuda:{rank % num_gpus}, `transcribe(model, audio).text`), `_evaluate_batch` —
    which the reference leaves unimplemented (:32-33) — recognises the whole batch through `transcribe_batch`."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from .k2.asr import TranscribeConfig as K2Config
        self.config = K2Config(verbose=False)

    @staticmethod
    def _load_model(device=None):
        from .k2.asr import load_model as lm
        return lm(device=device)

    @staticmethod
    def _transcribe(model, audio, config=None):
        from .k2.asr import transcribe as tr
        return tr(model, audio, config)

    @staticmethod
    def _transcribe_batch(model, audios, config=None):
        from .k2.asr import transcribe_batch as tb
        return tb(model, audios, config)
