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
    # no envelope - the node-by-node probe of tests/test_full_width_parity.py at THIS file's shape (2048 wide, 16 heads of 128, ring of 1 200):
    # both executors' rings hold the same pseudo-random BF16 rows, every layer is fed the same input at stream positions that put the merged attention in its
    # one-workgroup regime (60 / 159 live slots), at the split's threshold (161), deep in the split regime (900) and across the ring's wrap; clean nodes must
    # agree with the oracle to 2e-6, rounding flips are counted, a fused output may not move without a flip the per-node run shows.
    from test_full_width_parity import LayerProber
    cfg = temporal_cfg(1200)
    pr = LayerProber(cfg, taint_scale=5.0)   # (the test model's FFN is 1024 wide like the Depth transformer's: the probe's Depth bar below a counted flip)
    rng = np.random.default_rng(5)
    for layer in range(cfg.num_layers):
        for offset in (59, 158, 160, 161, 899, 1199, 1200, 2 * 1200 + 37):
            pr.fill_rings(0, layer, seed=100 + layer)
            x = (rng.standard_normal(cfg.dim) * 4).astype(np.float32)
            pr.one(0, layer, 0, x, offset, f"temporal layer {layer} offset {offset}")
    pr.finish("merged in_proj + attention shape (2048 wide, ring of 1 200)")
    # ... and inside the real cached graphs the merged launch is what runs (the bit-identity tests above compare it with the two-launch plan)
    m = hu.Model("hip", cfg, seed=0)
    hu.L.moshi_hot_fill_ring(m.m, 0, -1, 11, 1.0)
    hu.L.moshi_hot_set_context_fill(m.m, 900)
    rng = np.random.default_rng(5)
    for _ in range(3):
        m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
    assert m.stats().attention_folds_planned >= cfg.num_layers
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
