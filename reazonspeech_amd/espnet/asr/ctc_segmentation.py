"""CTC segmentation — the part of the third-party `ctc-segmentation` package the reference calls
(pkg/espnet-asr/src/ctc.py:4,60-75: `CtcSegmentationParameters(index_duration=..., char_list=...)`, `prepare_text`,
`ctc_segmentation`).

[UPSTREAM] `ctc-segmentation` (github.com/lumaku/ctc-segmentation; unpinned in pkg/espnet-asr/pyproject.toml:13) is not
installable here.  This module restates its published algorithm — Kürzinger et al., "CTC-Segmentation of Large Corpora for
German End-to-End Speech Recognition" (2020), §3 — with the package's conventions for the pieces the reference relies on:

  * ground truth  "#" + "·" + the characters of the text that are in `char_list` and not in `excluded_characters` + "·";
    `utt_begin_indices` = positions of the "·" before and after the utterance (the reference slices
    `timings[indices[0] + 1 : indices[1]]`: one time per kept character);
  * forward pass  k[t][c] = max(k[t-1][c] + max(p[t][blank], p[t][c]),  k[t-1][c-1] + p[t][c]);  staying in the preamble "#"
    costs nothing, so the alignment may start anywhere; k[0][c > 0] = max_prob (-1e10);
  * backtracking from the most probable frame of the last symbol; a character's time = (frame of its switch transition) x
    `index_duration`.
`p` is whatever the caller passes — the reference passes SOFTMAX PROBABILITIES where the package expects log-probabilities
(ctc.py:25-27 takes `ctc.softmax`); the arithmetic here is the same either way.  The package's windowing for very long inputs
(> `min_window_size` = 8000 frames) is not restated: the reference never aligns more than a 20 s window (~625 frames).
PARITY UNPINNED against the package itself.
"""
import numpy as np


class CtcSegmentationParameters:
    max_prob = -10000000000.0
    skip_prob = -10000000000.0
    min_window_size = 8000
    max_window_size = 100000
    index_duration = 0.025
    score_min_mean_over_L = 30
    space = "·"
    blank = 0
    replace_spaces_with_blanks = False
    blank_transition_cost_zero = False
    preamble_transition_cost_zero = True
    backtrack_from_max_t = False
    self_transition = "ε"
    start_of_ground_truth = "#"
    excluded_characters = ".,»«•❍·"
    tokenized_meta_symbol = "▁"
    char_list = None

    def __init__(self, **kwargs):
        self.set(**kwargs)

    def set(self, **kwargs):
        for key, val in kwargs.items():
            if not hasattr(self, key):
                raise ValueError(f"unknown CtcSegmentationParameters field {key!r}")
            setattr(self, key, val)


def prepare_text(config, text, char_list=None):
    """-> (ground_truth_mat int64 [len(ground truth)][max token length], utt_begin_indices)"""
    if char_list is not None:
        config.char_list = char_list
    blank = config.char_list[config.blank]
    ground_truth = config.start_of_ground_truth
    utt_begin_indices = []
    for utt in text:
        if not ground_truth.endswith(config.space):
            ground_truth += config.space
        utt_begin_indices.append(len(ground_truth) - 1)
        for char in utt:
            if char.isspace() and config.replace_spaces_with_blanks:
                if not ground_truth.endswith(config.space):
                    ground_truth += config.space
            elif char in config.char_list and char not in config.excluded_characters:
                ground_truth += char
    if not ground_truth.endswith(config.space):
        ground_truth += config.space
    utt_begin_indices.append(len(ground_truth) - 1)
    index = {}
    for i, c in enumerate(config.char_list):
        index.setdefault(c, i)
    max_char_len = max(len(c) for c in config.char_list)
    mat = np.full((len(ground_truth), max_char_len), -1, np.int64)
    for i in range(len(ground_truth)):
        for s in range(max_char_len):
            if i - s < 0:
                continue
            span = ground_truth[i - s:i + 1].replace(config.space, blank)
            if span in index:
                mat[i, s] = index[span]
    return mat, utt_begin_indices


def ctc_segmentation(config, lpz, ground_truth):
    """lpz float [T][V], ground_truth int64 [C][S] (prepare_text) -> (timings float64 [C], char_probs float64 [T], state_list)"""
    lpz = np.asarray(lpz, np.float32)
    gt = np.asarray(ground_truth, np.int64)
    T, C = lpz.shape[0], gt.shape[0]
    if C > T and config.skip_prob <= config.max_prob:
        raise AssertionError("Audio is shorter than text!")
    if T > config.min_window_size:
        raise NotImplementedError("windowed alignment of more than min_window_size frames is not restated (see the module docstring)")
    blank = config.blank
    neg = np.float32(config.max_prob)
    S = gt.shape[1]
    valid = gt >= 0                                              # [C][S]
    idx = np.where(valid, gt, 0)
    table = np.full((T, C), neg, np.float32)
    table[0, 0] = 0.0
    for t in range(1, T):
        prev = table[t - 1]
        pc = np.where(valid, lpz[t][idx], neg)                   # [C][S]: probability of the token ending at c with length s + 1
        best_char = pc.max(axis=1)                               # the stay transition may also repeat the character
        stay = prev + np.maximum(lpz[t, blank], best_char)
        if config.preamble_transition_cost_zero:
            stay[0] = prev[0]
        if config.blank_transition_cost_zero:
            stay = np.where(lpz[t, blank] >= best_char, prev, stay)
        switch = np.full((C,), neg, np.float32)
        for s in range(S):
            if s + 1 > C - 1:
                break
            cand = np.full((C,), neg, np.float32)
            cand[s + 1:] = prev[:C - s - 1] + pc[s + 1:, s]
            switch = np.maximum(switch, cand)
        table[t] = np.maximum(stay, switch)
    # backtracking from the most probable frame of the last symbol
    c = C - 1
    t = T - 1 if config.backtrack_from_max_t else int(np.argmax(table[:, c]))
    timings = np.zeros((C,), np.float64)
    char_probs = np.zeros((T,), np.float64)
    state_list = [""] * T
    while t != 0 or c != 0:
        if t == 0:
            raise IndexError("backtracking reached the first frame before the first symbol")
        min_s, min_delta, max_lpz = None, np.inf, config.max_prob
        for s in range(S):
            if gt[c, s] != -1 and c - 1 - s >= 0:
                switch_prob = float(lpz[t, gt[c, s]])
                est = float(table[t, c]) - float(table[t - 1, c - 1 - s])
                if abs(switch_prob - est) < min_delta:
                    min_delta, min_s = abs(switch_prob - est), s
                max_lpz = max(max_lpz, switch_prob)
        stay_prob = max(float(lpz[t, blank]), max_lpz)
        if c == 0 and config.preamble_transition_cost_zero:
            stay_prob = 0.0
        est_stay = float(table[t, c]) - float(table[t - 1, c])
        if min_s is not None and abs(stay_prob - est_stay) > min_delta:
            for s in range(min_s + 1):
                timings[c - s] = t * config.index_duration
            char_probs[t] = max_lpz
            state_list[t] = config.char_list[int(gt[c, min_s])]
            c -= 1 + min_s
            t -= 1
        else:
            char_probs[t] = stay_prob
            state_list[t] = config.self_transition
            t -= 1
    return timings, char_probs, state_list
