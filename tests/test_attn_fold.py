"""The Temporal layer's attention as the TAIL of its in_proj launch (inproj_attn_kernel, moshi.cpp_amd/csrc/hip_kernels_fused.hip; reference:
/root/reference/src/moshi/modules/transformer.h:449-576 - norm1, in_proj, RoPE, ring write, scaled-dot-product attention over the ring). The merged launch
performs the arithmetic of the two launches it replaces - matvec_q4k_kernel's Q8_K blocks, tile dots and row sums, the attention kernel's own body with
q / k / v handed over as tagged granules - so a plan with the merge (default) and one without (backend flag 512) must agree BIT FOR BIT at every live
length: the single-workgroup regime, the split regime, across the ring's wrap; and both must agree with the oracle."""
import ctypes as C

import numpy as np
import pytest

import hot_util as hu

pytestmark = pytest.mark.gpu


def temporal_cfg(context, dim=2048, heads=16, layers=2):
    # 2048 = 16 heads x 128: the narrowest width whose in_proj splits into whole 64-super-block tiles per (head part, q | k | v) segment
    cfg = hu.hot.tiny(hu.L, layers=layers, context=context)
    cfg.dim, cfg.num_heads = dim, heads
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    return cfg


def run(kind, cfg, steps, flags=0, fill=None, seed=3):
    m = hu.Model(kind, cfg, seed=0, flags=flags)
    if fill is not None:
        hu.L.moshi_hot_fill_ring(m.m, 0, -1, 11, 1.0)      # the same pseudo-random BF16 rows in every slot of both executors' rings
        hu.L.moshi_hot_set_context_fill(m.m, fill)
    rng = np.random.default_rng(seed)
    rec = []
    for _ in range(steps):
        r, txt, aud = m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
        rec.append((r, txt, aud, m.read("text_logits", cfg.text_card).copy(), m.read("stack_out", cfg.dim).copy()))
    st = m.stats() if kind == "hip" else None
    m.free()
    return rec, st


def assert_same(a, b, what):
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x[4], y[4]), f"{what} step {i}: Temporal stack output differs by {np.abs(x[4] - y[4]).max():.3e}"
        assert x[:3] == y[:3] and np.array_equal(x[3], y[3]), f"{what} step {i}: tokens / text logits differ"


@pytest.mark.parametrize("context,fill,steps", [(40, None, 90), (1200, 100, 5), (1200, 700, 5), (1200, 1195, 12), (3000, 2890, 4)])
def test_attention_in_the_tail_of_in_proj_equals_separate_launches_bit_for_bit(context, fill, steps):
    # ring of 40: wraps twice inside the run, one workgroup per head; rings of 1 200 / 3 000: the split geometry (a head over several of its 16 parts
    # beyond 160 live slots), incl. the wrap at 1 200
    cfg = temporal_cfg(context)
    tail, st = run("hip", cfg, steps, fill=fill)
    assert st.attention_folds_planned >= cfg.num_layers, f"the attention was not merged ({st.attention_folds_planned})"
    plain, st0 = run("hip", cfg, steps, flags=512, fill=fill)
    assert st0.attention_folds_planned == 0
    assert_same(plain, tail, f"context {context} fill {fill}")


def test_merged_attention_against_the_oracle():
    # teacher-forced (the oracle's tokens are written into the device model's delay ring after every step, so a near-tie in a Depth logit cannot fork the
    # two runs): the Temporal stack's output and the text logits at 60 and 900 live slots - single-workgroup and split regime of the merged attention
    cfg = temporal_cfg(1200)
    for fill in (60, 900):
        ms = {}
        for kind in ("oracle", "hip"):
            m = hu.Model(kind, cfg, seed=0)
            hu.L.moshi_hot_fill_ring(m.m, 0, -1, 11, 1.0)
            hu.L.moshi_hot_set_context_fill(m.m, fill)
            ms[kind] = m
        rng = np.random.default_rng(5)
        errs = []
        for i in range(6):
            ia = rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist()
            ms["oracle"].lm_step(ia)
            ms["hip"].lm_step(ia)
            traw, araw = C.c_int32(0), (C.c_int32 * 32)()
            hu.L.moshi_hot_last_raw_tokens(ms["oracle"].m, C.byref(traw), araw)
            hu.L.moshi_hot_force_last(ms["hip"].m, traw.value, araw)
            a, b = ms["oracle"].read("stack_out", cfg.dim), ms["hip"].read("stack_out", cfg.dim)
            errs.append((hu.rel_err(a, b), hu.rel_err(ms["oracle"].read("text_logits", cfg.text_card), ms["hip"].read("text_logits", cfg.text_card))))
        assert ms["hip"].stats().attention_folds_planned >= cfg.num_layers
        errs = np.array(errs)
        # summation noise, except where one Q8_K / BF16 value rounds the other way and taints what follows it through the ring (the per-layer "tainted" bar
        # of tests/test_full_width_parity.py; that the merge adds nothing to it is what the bit-identity tests establish)
        # (the node-by-node gate with counted rounding flips is tests/test_full_width_parity.py, which runs this plan; here: the envelope)
        assert errs[:, 0].max() < 3e-2, f"fill {fill}: stack output rel err per step {errs[:, 0]}"
        assert errs[:, 1].max() < 5e-2, f"fill {fill}: text logits rel err per step {errs[:, 1]}"
        for m in ms.values():
            m.free()


def test_merged_launch_replayed_from_a_hipgraph_many_times():
    # the launch tag comes from a counter in the workspace, bumped by the launch's last arriver: 60 replays of the captured launch against eager runs
    cfg = temporal_cfg(40, layers=1)
    tail, st = run("hip", cfg, 60)
    assert st.graph_replays > 0 and st.attention_folds_planned > 0
    plain, _ = run("hip", cfg, 60, flags=512 | 2)
    assert_same(plain, tail, "replayed merged launch vs eager launches")


def test_merge_at_the_moshika_width():
    # dim 4096 = 32 heads x 128, 8 parts per head, 3 x 16 rows x 16 super-blocks per workgroup: the benchmark's shape (2 layers)
    cfg = hu.hot.moshika(hu.L)
    cfg.num_layers = 2
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    for fill in (None, 2800):
        tail, st = run("hip", cfg, 4, fill=fill)
        assert st.attention_folds_planned >= 2
        plain, _ = run("hip", cfg, 4, flags=512, fill=fill)
        assert_same(plain, tail, f"moshika fill {fill}")


def test_narrow_models_keep_the_attention_launch():
    # 512-wide (4 heads): a (head part, segment) is not a whole tile - the planner must leave the attention a launch of its own
    cfg = temporal_cfg(40, dim=512, heads=4)
    dev, st = run("hip", cfg, 3)
    assert st.attention_folds_planned == 0
