"""The two searches `AVHubertForConditionalGeneration.generate` reaches through transformers' GenerationMixin with the README's
arguments (pkg/avsr/README.rst: `model.generate(**inputs, num_beams=5, max_new_tokens=256)`), as host control logic over the
device decoder: every step asks rs_avsr_decoder_step for the next-token logits of all hypothesis rows and decides here.

[UPSTREAM] transformers.generation.utils (<= 4.53.3 per pkg/avsr/pyproject.toml; the vectorised beam search introduced in 4.50):
  greedy  (`_sample`, do_sample False)  argmax per row; a row that emitted eos is fed / filled with pad_token_id; stops when all
          rows are done or after max_new_tokens
  beam    (`_beam_search`, early_stopping False, length_penalty 1.0)  per step the 2 K best (beam, token) continuations of a clip by
          accumulated log-probability; continuations that end (eos, or the length limit) and rank among the K best compete with
          score / generated_length ** length_penalty for the clip's K finished slots; the K best unfinished ones run on; a clip stops
          improving once its best running score / current_length ** length_penalty cannot beat its worst finished score, and the
          search ends when no clip can improve or nothing can continue.
The decoder prompt is one bos token (the model is not flagged is_encoder_decoder: generate() starts from bos_token_id)."""
import numpy as np
import torch


def greedy_search(dev, enc, padding_mask, max_new_tokens):
    cfg = dev.cfg
    B = enc.shape[0]
    dec = dev.decoding(enc, padding_mask, 1, 1 + max_new_tokens)
    ids = np.full((B, 1), cfg.bos_token_id, np.int64)
    unfinished = np.ones((B,), bool)
    for step in range(max_new_tokens):
        logits = dec.step(ids[:, -1], step)
        nxt = torch.argmax(logits, dim=-1).cpu().numpy()
        nxt = np.where(unfinished, nxt, cfg.pad_token_id)
        ids = np.concatenate([ids, nxt[:, None]], axis=1)
        unfinished &= nxt != cfg.eos_token_id
        if not unfinished.any():
            break
    return ids


def beam_search(dev, enc, padding_mask, num_beams, max_new_tokens, length_penalty=1.0):
    """-> (sequences int64 [B][L], scores float32 [B]): the best finished hypothesis of every clip"""
    cfg = dev.cfg
    B, K, V = enc.shape[0], int(num_beams), cfg.vocab_size
    max_len = 1 + max_new_tokens
    NEG = np.float32(-1.0e9)
    dec = dev.decoding(enc, padding_mask, K, max_len)
    run_seq = np.full((B, K, max_len), cfg.pad_token_id, np.int64)
    run_seq[:, :, 0] = cfg.bos_token_id
    run_score = np.zeros((B, K), np.float32)
    run_score[:, 1:] = NEG
    fin_seq, fin_score = run_seq.copy(), np.full((B, K), NEG, np.float32)
    fin_len = np.zeros((B, K), np.int64)
    is_fin = np.zeros((B, K), bool)
    can_improve = np.ones((B,), bool)
    src_rows = None
    rows_of = (np.arange(B) * K)[:, None]
    cur = 1
    while True:
        logits = dec.step(run_seq[:, :, cur - 1].reshape(-1), cur - 1, src_rows)
        logp = torch.log_softmax(logits.float(), dim=-1).view(B, K, V) + torch.from_numpy(run_score).to(logits.device)[:, :, None]
        top_lp, top_idx = torch.topk(logp.view(B, K * V), k=2 * K)
        top_lp, top_idx = top_lp.cpu().numpy(), top_idx.cpu().numpy()
        parent, token = top_idx // V, top_idx % V
        cand = np.take_along_axis(run_seq, parent[:, :, None], axis=1)
        cand[:, :, cur] = token
        ends = (token == cfg.eos_token_id) | (cur + 1 >= max_len)
        # the K best continuations that go on
        lp_run = top_lp + ends.astype(np.float32) * NEG
        keep = np.argsort(-lp_run, axis=1, kind="stable")[:, :K]
        run_seq = np.take_along_axis(cand, keep[:, :, None], axis=1)
        run_score = np.take_along_axis(lp_run, keep, axis=1)
        src_rows = (rows_of + np.take_along_axis(parent, keep, axis=1)).reshape(-1)
        # finished hypotheses: only a continuation ranked among the K best may finish
        just = ends.copy()
        just[:, K:] = False
        lp_fin = top_lp / np.float32(float(cur) ** length_penalty)
        lp_fin = lp_fin + (~can_improve)[:, None].astype(np.float32) * NEG + (~just).astype(np.float32) * NEG
        m_score = np.concatenate([fin_score, lp_fin], axis=1)
        best = np.argsort(-m_score, axis=1, kind="stable")[:, :K]
        fin_seq = np.take_along_axis(np.concatenate([fin_seq, cand], axis=1), best[:, :, None], axis=1)
        fin_len = np.take_along_axis(np.concatenate([fin_len, np.full((B, 2 * K), cur + 1, np.int64)], axis=1), best, axis=1)
        is_fin = np.take_along_axis(np.concatenate([is_fin, just], axis=1), best, axis=1)
        fin_score = np.take_along_axis(m_score, best, axis=1)
        cur += 1
        best_running = run_score[:, 0] / np.float32(float(cur - 1) ** length_penalty)
        worst_finished = np.where(is_fin, fin_score.min(axis=1, keepdims=True), NEG)
        can_improve &= (best_running[:, None] > worst_finished).any(axis=1)
        if not (can_improve.any() and not ends.all()):
            break
    n = int(fin_len[:, 0].max())
    return fin_seq[:, 0, :n], fin_score[:, 0]
