"""Flip audit of greedy RNN-T decisions.  TEST INFRASTRUCTURE (oracle/__init__.py).

Greedy ids of the HIP path against the END-TO-END fp32 oracle are not an identity: the two sides feed the (exact,
fixed-order float32) decode loop with joint-encoder tensors that differ by the encoder's bf16 noise, and an argmax
whose top two logits are closer than that noise can move them may flip.  This module turns "token agreement 0.94" into
"exact except provable near-ties" ([UPSTREAM] GreedyBatchedRNNTInfer; restated at oracle/model.py: greedy_torch):

  * it walks the HIP hypothesis' own decision path (the tokens it emitted at each frame, then the blank that advanced
    the frame) with the prediction network in float64, so at every decision point both sides share the same history
    (same `g`) and differ ONLY in the encoder projection row f[t];
  * at each point it evaluates the joint on the oracle's row (z_ref) and on the HIP row (z_hip).  Where the two argmaxes
    differ ("local flip"), ReLU being 1-Lipschitz gives
        z_ref[k_ref] - z_ref[k_hip]  <=  (|w[k_ref]| + |w[k_hip]|) * |f_ref[t] - f_hip[t]|_2
    so a flip is only possible when the oracle's own margin between the two candidates is below that bound;
  * every difference between the two id sequences starts at a local flip: an utterance with no local flip along its
    path must have ids identical to the oracle's end-to-end ids (checked by the callers).
"""
import numpy as np


def _decoder_view(cfg, sd):
    """the decoder / joint tensors under the NeMo key names this module reads, whatever the model family: an ESPnet2 state
    dict ([UPSTREAM] TransducerDecoder: one LSTM module per layer, JointNetwork: lin_dec without a bias, tanh) is re-keyed.
    The walk itself is family-independent — tanh is 1-Lipschitz like ReLU, so the flip bound is the same, and ESPnet's
    greedy_search (one symbol per frame) is the max_symbols = 1 case of the decision lists."""
    if not getattr(cfg, "espnet", False):
        return sd
    import torch
    out = {"decoder.prediction.embed.weight": sd["decoder.embed.weight"],
           "joint.pred.weight": sd["joint_network.lin_dec.weight"],
           "joint.pred.bias": torch.zeros((cfg.joint_hidden,), dtype=torch.float32),
           "joint.joint_net.2.weight": sd["joint_network.lin_out.weight"],
           "joint.joint_net.2.bias": sd["joint_network.lin_out.bias"]}
    for l in range(cfg.pred_layers):
        for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            out[f"decoder.prediction.dec_rnn.lstm.{nm}_l{l}"] = sd[f"decoder.decoder.{l}.{nm}_l0"]
    return out


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


class _PredNet:
    """Embedding -> LSTM (gate order i, f, g, o) -> Linear, float64 ([UPSTREAM] RNNTDecoder.predict, RNNTJoint.pred)"""

    def __init__(self, cfg, sd):
        P = "decoder.prediction.dec_rnn.lstm."
        f64 = lambda k: sd[k].detach().cpu().numpy().astype(np.float64)  # noqa: E731
        self.cfg = cfg
        self.emb = f64("decoder.prediction.embed.weight")
        self.w = [np.concatenate([f64(P + f"weight_ih_l{l}"), f64(P + f"weight_hh_l{l}")], axis=1) for l in range(cfg.pred_layers)]
        self.b = [f64(P + f"bias_ih_l{l}") + f64(P + f"bias_hh_l{l}") for l in range(cfg.pred_layers)]
        self.wp, self.bp = f64("joint.pred.weight"), f64("joint.pred.bias")
        H = cfg.pred_hidden
        self.h = [np.zeros(H) for _ in range(cfg.pred_layers)]
        self.c = [np.zeros(H) for _ in range(cfg.pred_layers)]

    def step(self, token):
        H = self.cfg.pred_hidden
        x = self.emb[token]
        for l in range(self.cfg.pred_layers):
            z = self.w[l] @ np.concatenate([x, self.h[l]]) + self.b[l]
            i, f, g, o = _sigmoid(z[:H]), _sigmoid(z[H:2 * H]), np.tanh(z[2 * H:3 * H]), _sigmoid(z[3 * H:])
            self.c[l] = f * self.c[l] + i * g
            self.h[l] = o * np.tanh(self.c[l])
            x = self.h[l]
        return self.wp @ x + self.bp


def flip_audit(cfg, sd, f_ref, f_hip, enc_len, hip_ids, hip_frames):
    """f_ref / f_hip: float arrays [>= enc_len, J] (oracle / HIP joint-encoder projection of ONE utterance);
    hip_ids / hip_frames: the HIP hypothesis.  Returns a dict:
      decisions      decision points walked (every emitted token and every frame-advancing blank)
      flips          list of {frame, k_ref, k_hip, margin_ref, bound, delta_f} for local flips
      margins        float64 array: the oracle's top-1 minus top-2 logit at every decision point
      path_ok        the float64 argmax on the HIP rows reproduces the HIP hypothesis at every point whose float64
                     top-2 margin exceeds 1e-3 (sanity of the walk itself; the bit-exact statement is the C oracle's)
    """
    sd = _decoder_view(cfg, sd)
    act = np.tanh if getattr(cfg, "espnet", False) else (lambda a: np.maximum(a, 0.0))
    f_ref = np.asarray(f_ref, np.float64)
    f_hip = np.asarray(f_hip, np.float64)
    wo = sd["joint.joint_net.2.weight"].detach().cpu().numpy().astype(np.float64)
    bo = sd["joint.joint_net.2.bias"].detach().cpu().numpy().astype(np.float64)
    wnorm = np.linalg.norm(wo, axis=1)
    blank = cfg.blank_id
    by_frame = {}
    for k, t in zip(hip_ids, hip_frames):
        by_frame.setdefault(int(t), []).append(int(k))
    pred = _PredNet(cfg, sd)
    g = pred.step(blank)
    flips, margins, path_ok = [], [], True
    n_dec = 0
    for t in range(int(enc_len)):
        toks = by_frame.get(t, [])
        path = toks + ([blank] if len(toks) < cfg.max_symbols else [])
        for k in path:
            z_ref = wo @ act(f_ref[t] + g) + bo
            z_hip = wo @ act(f_hip[t] + g) + bo
            n_dec += 1
            top = np.argpartition(z_ref, -2)[-2:]
            margins.append(float(abs(z_ref[top[1]] - z_ref[top[0]])))
            k_hip_f64 = int(np.argmax(z_hip))
            if k_hip_f64 != k:
                srt = np.sort(z_hip)
                if srt[-1] - z_hip[k] > 1e-3:
                    path_ok = False
            k_ref = int(np.argmax(z_ref))
            if k_ref != k:
                df = float(np.linalg.norm(f_ref[t] - f_hip[t]))
                flips.append({"frame": t, "k_ref": k_ref, "k_hip": k, "margin_ref": float(z_ref[k_ref] - z_ref[k]),
                              "bound": float((wnorm[k_ref] + wnorm[k]) * df), "delta_f": df})
            if k != blank:
                g = pred.step(k)
    return {"decisions": n_dec, "flips": flips, "margins": np.asarray(margins), "path_ok": path_ok}


def summarize(audits, ids_equal):
    """fold per-utterance audits into the numbers the parity reports carry.  `ids_equal[b]`: HIP ids == oracle
    end-to-end ids.  The implication `no local flip => ids equal` is returned as `explained` (must be all True)."""
    margins = np.concatenate([a["margins"] for a in audits]) if audits else np.zeros(0)
    flips = [f for a in audits for f in a["flips"]]
    explained = [bool(eq) or len(a["flips"]) > 0 for a, eq in zip(audits, ids_equal)]
    out = {"decisions": int(sum(a["decisions"] for a in audits)), "local_flips": len(flips),
           "utterances_without_flip": int(sum(len(a["flips"]) == 0 for a in audits)),
           "every_id_difference_starts_at_a_flip": all(explained),
           "walk_reproduces_hip_path": all(a["path_ok"] for a in audits)}
    if len(margins):
        out["oracle_margin_median"] = float(np.median(margins))
        out["oracle_margin_p05"] = float(np.percentile(margins, 5))
    if flips:
        fm = np.asarray([f["margin_ref"] for f in flips])
        out["flip_margin_max"] = float(fm.max())
        out["flip_margin_over_bound_max"] = float(max(f["margin_ref"] / max(f["bound"], 1e-30) for f in flips))
        out["flip_margin_percentile_of_all_margins_max"] = float((margins < fm.max()).mean() * 100.0)
    return out


def flip_audit_batch(cfg, sd, f_ref, f_hip, enc_lens, hyp_ids, hyp_frames, device="cpu"):
    """`flip_audit` for a whole batch at once, in torch float64 on `device` (a GPU box audits all 256 rows of the benchmark
    batch in a second or two; the per-row numpy walk above takes about half a second per row).  Rows advance in lockstep
    over their own decision lists; the arithmetic per row is the same as in `flip_audit`.

      f_ref, f_hip   [B, T, J] tensors (any float dtype / device): reference and audited joint-encoder projections
      enc_lens       B ints;  hyp_ids / hyp_frames: the AUDITED side's hypotheses (lists of lists)
    -> list of per-row dicts with the keys of `flip_audit` (margins as numpy float64 arrays)."""
    import torch
    sd = _decoder_view(cfg, sd)
    act = torch.tanh if getattr(cfg, "espnet", False) else torch.relu
    dev = torch.device(device)
    f64 = lambda t: t.detach().to(device=dev, dtype=torch.float64)  # noqa: E731
    B = len(enc_lens)
    blank, H, L = cfg.blank_id, cfg.pred_hidden, cfg.pred_layers
    P = "decoder.prediction.dec_rnn.lstm."
    emb = f64(sd["decoder.prediction.embed.weight"])
    w = [torch.cat([f64(sd[P + f"weight_ih_l{l}"]), f64(sd[P + f"weight_hh_l{l}"])], dim=1) for l in range(L)]
    bsum = [f64(sd[P + f"bias_ih_l{l}"]) + f64(sd[P + f"bias_hh_l{l}"]) for l in range(L)]
    wp, bp = f64(sd["joint.pred.weight"]), f64(sd["joint.pred.bias"])
    wo, bo = f64(sd["joint.joint_net.2.weight"]), f64(sd["joint.joint_net.2.bias"])
    wnorm = wo.norm(dim=1)
    fr, fh = f64(f_ref), f64(f_hip)
    # decision lists: at frame t the tokens emitted there, then the blank that advanced the frame
    dec_t, dec_k = [], []
    for b in range(B):
        by_frame = {}
        for k, t in zip(hyp_ids[b], hyp_frames[b]):
            by_frame.setdefault(int(t), []).append(int(k))
        ts, ks = [], []
        for t in range(int(enc_lens[b])):
            toks = by_frame.get(t, [])
            for k in toks + ([blank] if len(toks) < cfg.max_symbols else []):
                ts.append(t); ks.append(k)
        dec_t.append(ts); dec_k.append(ks)
    D = max((len(x) for x in dec_t), default=0)
    T = torch.zeros((B, max(D, 1)), dtype=torch.long)
    K = torch.full((B, max(D, 1)), blank, dtype=torch.long)
    V = torch.zeros((B, max(D, 1)), dtype=torch.bool)
    for b in range(B):
        n = len(dec_t[b])
        T[b, :n] = torch.tensor(dec_t[b], dtype=torch.long)
        K[b, :n] = torch.tensor(dec_k[b], dtype=torch.long)
        V[b, :n] = True
    T, K, V = T.to(dev), K.to(dev), V.to(dev)
    h = [torch.zeros((B, H), dtype=torch.float64, device=dev) for _ in range(L)]
    c = [torch.zeros((B, H), dtype=torch.float64, device=dev) for _ in range(L)]

    def pred_step(tokens, rows):
        """advance the prediction network of `rows` (bool [B]) with `tokens` [B]; -> g [B, J] for those rows"""
        x = emb[tokens]
        for l in range(L):
            z = torch.cat([x, h[l]], dim=1) @ w[l].t() + bsum[l]
            i, f, g_, o = torch.sigmoid(z[:, :H]), torch.sigmoid(z[:, H:2 * H]), torch.tanh(z[:, 2 * H:3 * H]), torch.sigmoid(z[:, 3 * H:])
            cn = f * c[l] + i * g_
            hn = o * torch.tanh(cn)
            c[l] = torch.where(rows[:, None], cn, c[l])
            h[l] = torch.where(rows[:, None], hn, h[l])
            x = h[l]
        return x @ wp.t() + bp

    every = torch.ones((B,), dtype=torch.bool, device=dev)
    g = pred_step(torch.full((B,), blank, dtype=torch.long, device=dev), every)
    rows_idx = torch.arange(B, device=dev)
    margins = torch.zeros((B, max(D, 1)), dtype=torch.float64, device=dev)
    k_ref_all = torch.zeros((B, max(D, 1)), dtype=torch.long, device=dev)
    m_flip = torch.zeros((B, max(D, 1)), dtype=torch.float64, device=dev)
    df_all = torch.zeros((B, max(D, 1)), dtype=torch.float64, device=dev)
    path_bad = torch.zeros((B,), dtype=torch.bool, device=dev)
    for d in range(D):
        t, k, v = T[:, d], K[:, d], V[:, d]
        fr_t, fh_t = fr[rows_idx, t], fh[rows_idx, t]
        z_ref = act(fr_t + g) @ wo.t() + bo
        z_hip = act(fh_t + g) @ wo.t() + bo
        top2 = z_ref.topk(2, dim=1).values
        margins[:, d] = top2[:, 0] - top2[:, 1]
        k_ref = z_ref.argmax(dim=1)
        k_ref_all[:, d] = k_ref
        m_flip[:, d] = z_ref[rows_idx, k_ref] - z_ref[rows_idx, k]
        df_all[:, d] = (fr_t - fh_t).norm(dim=1)
        path_bad |= v & ((z_hip.max(dim=1).values - z_hip[rows_idx, k]) > 1e-3)
        emit = v & (k != blank)
        if bool(emit.any()):
            g_new = pred_step(k, emit)
            g = torch.where(emit[:, None], g_new, g)
    margins, k_ref_all, m_flip, df_all = margins.cpu().numpy(), k_ref_all.cpu().numpy(), m_flip.cpu().numpy(), df_all.cpu().numpy()
    wn = wnorm.cpu().numpy()
    path_bad = path_bad.cpu().numpy()
    out = []
    for b in range(B):
        n = len(dec_t[b])
        flips = []
        for d in range(n):
            if int(k_ref_all[b, d]) != dec_k[b][d]:
                kr, kh = int(k_ref_all[b, d]), dec_k[b][d]
                flips.append({"frame": dec_t[b][d], "k_ref": kr, "k_hip": kh, "margin_ref": float(m_flip[b, d]),
                              "bound": float((wn[kr] + wn[kh]) * df_all[b, d]), "delta_f": float(df_all[b, d])})
        out.append({"decisions": n, "flips": flips, "margins": margins[b, :n].astype(np.float64), "path_ok": not bool(path_bad[b])})
    return out


def flip_audit_batch_k2(cfg, sd, f_ref, f_hip, enc_lens, hyp_ids, hyp_frames, device="cpu"):
    """`flip_audit_batch` for the Zipformer family ([UPSTREAM] sherpa-onnx greedy search over icefall's STATELESS decoder;
    restated at oracle/zipformer.py: greedy_search): one decision per frame, the context is the last `context_size` emitted
    tokens, and blank and `<unk>` are one decision class ("nothing emitted": neither is recorded nor enters the context).
    The walk follows the audited side's own hypothesis with the decoder in float64, so at every frame both sides share the
    context and differ only in the encoder-projection row; tanh is 1-Lipschitz, hence the same bound as for the LSTM families:
        z_ref[k_ref] - z_ref[k_hip]  <=  (|w[k_ref]| + |w[k_hip]|) |f_ref[t] - f_hip[t]|_2
    where, for a frame without an emission, k_hip is the better of (blank, unk) on the audited row.
    -> list of per-row dicts with the keys of `flip_audit`."""
    import torch
    dev = torch.device(device)
    f64 = lambda t: t.detach().to(device=dev, dtype=torch.float64)  # noqa: E731
    B = len(enc_lens)
    blank, unk, C = cfg.blank_id, cfg.unk_id, cfg.context_size
    emb, cw = f64(sd["decoder.embedding.weight"]), f64(sd["decoder.conv.weight"])            # [V][D], [D][4][C]
    wp, bp = f64(sd["joiner.decoder_proj.weight"]), f64(sd["joiner.decoder_proj.bias"])
    wo, bo = f64(sd["joiner.output_linear.weight"]), f64(sd["joiner.output_linear.bias"])
    Dd = emb.shape[1]
    wnorm = wo.norm(dim=1)
    fr, fh = f64(f_ref), f64(f_hip)
    Tm = int(max(enc_lens)) if B else 0
    emitted = torch.full((B, max(Tm, 1)), -1, dtype=torch.long)
    for b in range(B):
        for k, t in zip(hyp_ids[b], hyp_frames[b]):
            emitted[b, int(t)] = int(k)
    emitted = emitted.to(dev)
    lens = torch.as_tensor(list(enc_lens), dtype=torch.long, device=dev)
    hist = torch.full((B, C), -1, dtype=torch.long, device=dev)
    hist[:, -1] = blank

    def dec(h):
        e = emb[h.clamp(min=0)] * (h >= 0).unsqueeze(-1)                                         # [B][C][D]
        e = e.permute(0, 2, 1).reshape(B, Dd // 4, 1, 4, C).expand(B, Dd // 4, 4, 4, C).reshape(B, Dd, 4, C)
        return torch.relu((e * cw[None]).sum(dim=(2, 3))) @ wp.t() + bp

    g = dec(hist)
    none = [blank] + ([unk] if unk >= 0 else [])
    margins = torch.zeros((B, max(Tm, 1)), dtype=torch.float64, device=dev)
    k_ref_all = torch.zeros((B, max(Tm, 1)), dtype=torch.long, device=dev)
    k_hip_all = torch.zeros((B, max(Tm, 1)), dtype=torch.long, device=dev)
    m_flip = torch.zeros((B, max(Tm, 1)), dtype=torch.float64, device=dev)
    df_all = torch.zeros((B, max(Tm, 1)), dtype=torch.float64, device=dev)
    path_bad = torch.zeros((B,), dtype=torch.bool, device=dev)
    rows = torch.arange(B, device=dev)
    for t in range(Tm):
        v = lens > t
        z_ref = torch.tanh(fr[:, t] + g) @ wo.t() + bo
        z_hip = torch.tanh(fh[:, t] + g) @ wo.t() + bo
        top2 = z_ref.topk(2, dim=1).values
        margins[:, t] = top2[:, 0] - top2[:, 1]
        best_none = torch.tensor(none, device=dev)[z_hip[:, none].argmax(dim=1)]
        k_hip = torch.where(emitted[:, t] >= 0, emitted[:, t], best_none)
        k_ref = z_ref.argmax(dim=1)
        k_ref_all[:, t], k_hip_all[:, t] = k_ref, k_hip
        m_flip[:, t] = z_ref[rows, k_ref] - z_ref[rows, k_hip]
        df_all[:, t] = (fr[:, t] - fh[:, t]).norm(dim=1)
        path_bad |= v & ((z_hip.max(dim=1).values - z_hip[rows, k_hip]) > 1e-3)
        emit = v & (emitted[:, t] >= 0)
        if bool(emit.any()):
            hist = torch.where(emit[:, None], torch.cat([hist[:, 1:], emitted[:, t:t + 1]], dim=1), hist)
            g = torch.where(emit[:, None], dec(hist), g)
    margins, k_ref_all, k_hip_all = margins.cpu().numpy(), k_ref_all.cpu().numpy(), k_hip_all.cpu().numpy()
    m_flip, df_all, wn, path_bad = m_flip.cpu().numpy(), df_all.cpu().numpy(), wnorm.cpu().numpy(), path_bad.cpu().numpy()
    cls = lambda k: -1 if k in none else k          # noqa: E731
    out = []
    for b in range(B):
        n = int(enc_lens[b])
        flips = []
        for t in range(n):
            kr, kh = int(k_ref_all[b, t]), int(k_hip_all[b, t])
            if cls(kr) != cls(kh):
                flips.append({"frame": t, "k_ref": kr, "k_hip": kh, "margin_ref": float(m_flip[b, t]),
                              "bound": float((wn[kr] + wn[kh]) * df_all[b, t]), "delta_f": float(df_all[b, t])})
        out.append({"decisions": n, "flips": flips, "margins": margins[b, :n].astype(np.float64), "path_ok": not bool(path_bad[b])})
    return out
