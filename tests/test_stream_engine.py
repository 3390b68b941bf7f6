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


def test_streamed_layers_agree_with_the_oracle():
    cfg = temporal_at_real_width(layers=2)
    steps = 3
    streamed, st = run("hip", cfg, steps)
    assert st.streamed_matvecs_planned > 0
    ref, _ = run("oracle", cfg, steps)
    compared = 0
    for i, (a, b) in enumerate(zip(ref, streamed)):
        # logits first (a greedy token can only differ where the oracle's own two best logits are a near-tie; the trajectories part there)
        assert hu.rel_err(a[3], b[3]) < 1e-3, f"step {i}: text logits {hu.rel_err(a[3], b[3]):.3e}"
        assert a[1] == b[1], f"step {i}: text token {a[1]} vs {b[1]}"
        same = True
        for k in range(cfg.dep_q):
            assert hu.rel_err(a[4][k], b[4][k]) < 1e-3, f"step {i} depth {k}: logits {hu.rel_err(a[4][k], b[4][k]):.3e}"
            if a[2][k] != b[2][k]:
                top = np.sort(a[4][k])[-2:]
                assert top[1] - top[0] < 1e-3 * np.abs(a[4][k]).max(), f"step {i} depth {k}: token {a[2][k]} vs {b[2][k]} without a near-tie ({top})"
                same = False
                break
        compared += 1
        if not same:
            break
    assert compared >= 1
