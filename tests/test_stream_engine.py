"""The persistent stream engine (moshi.cpp_amd/csrc/hip_stream.hip): the large Q4_K mat-vecs of a Temporal layer between two attention launches -
out_proj + residual, norm2 + gated linear_in, linear_out + residual, the next layer's norm1 + in_proj
(/root/reference/src/moshi/modules/transformer.h:300-420) - executed by one launch. It performs the arithmetic of the one-launch-per-mat-vec
plan in the same order, so the two plans (backend flag 64 switches the stream launches off) must agree BIT FOR BIT, and both with the oracle."""
import numpy as np
import pytest

import hot_util as hu

pytestmark = pytest.mark.gpu


def temporal_at_real_width(which="moshika", layers=3, context=64):
    cfg = hu.hot.moshika(hu.L) if which == "moshika" else hu.hot.personaplex(hu.L)
    cfg.num_layers, cfg.context = layers, context
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    return cfg


def run(kind, cfg, steps, flags=128, seed=3):
    m = hu.Model(kind, cfg, seed=0, flags=flags)
    rng = np.random.default_rng(seed)
    rec = []
    n_in = cfg.n_q - cfg.io_dep_q
    for _ in range(steps):
        ia = rng.integers(0, cfg.card, n_in).tolist()
        r, txt, aud = m.lm_step(ia)
        rec.append((r, txt, aud, m.read("text_logits", cfg.text_card).copy(), [m.read(f"dep_logits{k}", cfg.card).copy() for k in range(cfg.dep_q)]))
    st = m.stats() if kind == "hip" else None
    m.free()
    return rec, st


def assert_bit_identical(a, b, what):
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[:3] == y[:3], f"{what} step {i}: tokens {x[:3]} vs {y[:3]}"
        assert np.array_equal(x[3], y[3]), f"{what} step {i}: text logits differ by {np.abs(x[3] - y[3]).max():.3e}"
        for k, (u, v) in enumerate(zip(x[4], y[4])):
            assert np.array_equal(u, v), f"{what} step {i} depth {k}: logits differ by {np.abs(u - v).max():.3e}"


@pytest.mark.parametrize("which", ["moshika", "personaplex"])
def test_streamed_temporal_layers_equal_one_launch_per_matvec_bit_for_bit(which):
    cfg = temporal_at_real_width(which)
    steps = 6
    streamed, st = run("hip", cfg, steps)
    # layer 0's in_proj runs alone; then {out_proj, linear_in, linear_out, next in_proj} per layer; the last run ends in front of the text head
    assert st.streamed_matvecs_planned == 4 * (cfg.num_layers - 1) + 3, f"{st.streamed_matvecs_planned} Temporal mat-vecs in stream launches"
    plain, st0 = run("hip", cfg, steps, flags=64)
    assert st0.streamed_matvecs_planned == 0
    assert_bit_identical(plain, streamed, which)


def test_streamed_layers_replayed_from_a_hipgraph_stay_identical_to_eager_launches():
    # hand-off tags derive from a launch counter kept on the device: 30 replays of the captured launches against 30 eager unstreamed runs
    cfg = temporal_at_real_width(layers=2)
    streamed, st = run("hip", cfg, 30)
    assert st.graph_replays > 0 and st.streamed_matvecs_planned > 0
    plain, _ = run("hip", cfg, 30, flags=64 | 2)
    assert_bit_identical(plain, streamed, "replayed stream launches vs eager launches")


def test_streamed_layers_agree_with_the_oracle_on_the_first_frame():
    # Random full-width weights amplify a Q8_K rounding flip by ~1e-2 per layer (tests/test_full_width_parity.py pins that node by node, with
    # contractive weights, for the one-launch-per-mat-vec plan - which the tests above show to be this plan bit for bit); this is the end-to-end
    # guard on top: the first frame's logits within 5e-2 of the oracle's, in relative L2.
    cfg = temporal_at_real_width(layers=2)
    streamed, st = run("hip", cfg, 1)
    assert st.streamed_matvecs_planned > 0
    ref, _ = run("oracle", cfg, 1)
    a, b = ref[0], streamed[0]
    assert hu.rel_err(a[3], b[3]) < 5e-2, f"text logits {hu.rel_err(a[3], b[3]):.3e}"
    assert hu.rel_err(a[4][0], b[4][0]) < 5e-2, f"first Depth logits {hu.rel_err(a[4][0], b[4][0]):.3e}"
