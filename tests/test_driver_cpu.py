"""CPU: the host-side hot-path driver (include/moshi_hot.h) on the host device with the oracle attached —
frame protocol, delay ring, determinism, and the bias-mask lookup semantics incl. the T = 2 quirk."""
import numpy as np
import pytest

import ggml_util as gu
import hot_util as hu
from ggml_util import F32


def test_delay_ring_first_frame_produces_nothing_then_frames_flow():
    cfg = hu.hot.tiny(hu.L, layers=1)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    m = hu.Model("oracle", cfg)
    rng = np.random.default_rng(0)
    outs = [m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist()) for _ in range(4)]
    m.free()
    assert outs[0][0] == 0                      # offset <= max_delay (lm.h:950)
    assert all(o[0] == 1 for o in outs[1:])
    assert all(0 <= t < cfg.card for o in outs[1:] for t in o[2]) and all(0 <= o[1] < cfg.text_card for o in outs[1:])


def test_driver_is_deterministic_in_seed():
    cfg = hu.hot.tiny(hu.L, layers=1)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    runs = []
    for seed in (0, 0, 1):
        m = hu.Model("oracle", cfg, seed=seed)
        runs.append([m.lm_step([1, 2, 3]) for _ in range(3)])
        m.free()
    assert runs[0] == runs[1] and runs[0] != runs[2]


def test_codec_round_trip_runs_and_is_finite():
    cfg = hu.hot.tiny(hu.L)
    cfg.enable_lm = 0
    m = hu.Model("oracle", cfg)
    rng = np.random.default_rng(1)
    for _ in range(2):
        codes = m.mimi_encode(rng.standard_normal(1920).astype(np.float32) * 0.1)
        assert len(codes) == cfg.mimi_n_q and all(0 <= c < cfg.mimi_codebook_size for c in codes)
        pcm = m.mimi_decode(codes)
        assert pcm.shape == (1920,) and np.isfinite(pcm).all() and np.abs(pcm).max() > 0
    m.free()


def visible_slots(C, T, offset):
    """Which ring slots each of the T query rows may attend at stream position `offset`, read from the graph the
    driver/reference build: cont(view_2d(pattern, C, T, nb1, col)) (torch.h:170-223), executed by the oracle."""
    start, width = 2 * C - T, 3 * C - T
    pat = np.zeros((T, width), np.float32)
    for j in range(T):
        pat[j, start + 1 + j:] = -np.inf
        for i in range(T - j - 1):
            pat[j, C - 1 - i] = -np.inf
    col = start - offset if offset <= C else C - (offset % C)

    def build(g):
        p = g.input(pat)
        v = g.view_2d(p, C, T, p.contents.nb[1], col * 4)
        return [g.cont(v)]
    (m,), _ = gu.run_graph("oracle", build)
    return [set(np.nonzero(np.isfinite(m.reshape(T, C)[j]))[0].tolist()) for j in range(T)]


def test_mask_t1_is_filled_prefix_then_everything():
    C = 12
    for off in range(0, 40):
        vis = visible_slots(C, 1, off)[0]
        assert vis == (set(range(off + 1)) if off < C else set(range(C))), off


def test_mask_t2_causal_before_wrap_and_reference_quirk_after():
    C = 10
    for off in range(0, C - 1, 2):              # chunk occupies slots off, off+1
        v0, v1 = visible_slots(C, 2, off)
        assert v0 == set(range(off + 1)) and v1 == set(range(off + 2)), off
    # once offset > C the intra-chunk mask is shifted two columns left: query row 0 masks the slot
    # before the chunk and sees the slot after it. Parity means reproducing this, not fixing it.
    off = C + 4
    s = off % C
    v0, v1 = visible_slots(C, 2, off)
    assert v1 == set(range(C))
    assert v0 == set(range(C)) - {(s - 1) % C}
