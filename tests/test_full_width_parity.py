"""-m gpu: parity AT THE BENCHMARK SHAPE (moshika-7B q4_k: dim 4096, 32 heads, 32 + 6 layers, ffn 11264 / 2816, ring 3000 / 8), MI355X
backend vs the CPU oracle, with tight assertions (north_star: bit-exact greedy ids, logits 1e-3 / one quantiser step).

Three complementary angles, because ggml's arithmetic is chaotic at this width on random weights:
  1. every Temporal and Depth LAYER, teacher-forced with the oracle's layer input, compared NODE BY NODE (tests/parity_probe.py): nodes
     agree to 2e-6 of max unless a counted BF16 / Q8_K rounding flip - a tie by construction - sits upstream; per-node kernels AND the
     fused kernels bench.py times;
  2. a CONTRACTIVE synthetic-weight variant (update_scale < 1, include/moshi_hot.h): same shapes, types and bytes, but rounding flips no
     longer compound, so >= 32 FREE-RUNNING greedy frames are asserted bit-exact with logits within 1e-3;
  3. BASELINE configs[2]'s codec leg: the 32-level Mimi ENCODER over 133 frames (10 s + 8 tail frames, tools/moshi-stt.cpp:549-577), which
     takes the encoder transformer's offset past 250 = across the T = 2 mask quirk; codes bit-exact.
"""
import numpy as np
import pytest

import hot_util as hu
import parity_probe as pp

pytestmark = pytest.mark.gpu
L = hu.L


def lm_only(cfg):
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    return cfg


def test_every_full_width_layer_node_by_node_teacher_forced():
    cfg = lm_only(hu.hot.moshika(L))
    ref = hu.Model("oracle", cfg, seed=0)
    dev = hu.Model("hip", cfg, seed=0)
    rng = np.random.default_rng(11)
    tot = {"nodes": 0, "clean": 0, "tainted": 0, "flips": 0, "sites": 0, "hidden": 0, "fused_nodes": 0, "fused_clean": 0}
    worst_clean = worst_tainted = 0.0
    flipped_sites = []

    def one(which, layer, ws, x, offset, where):
        nonlocal worst_clean, worst_tainted
        a, ya = pp.probe(ref, which, layer, ws, x, offset)
        L.ggml_backend_mi355x_set_flags(dev.be, 1 | 2 | 4)          # one generic kernel per node: every node is visible
        b, _ = pp.probe(dev, which, layer, ws, x, offset)
        st = pp.compare_layer(a, b, where + " per-node")
        L.ggml_backend_mi355x_set_flags(dev.be, 0)                  # the fused kernels of the benchmark (interior nodes never materialise)
        c, yc = pp.probe(dev, which, layer, ws, x, offset)
        sf = pp.compare_layer(a, c, where + " fused", taint_in=st["taint"], hidden_flips=True)
        assert sf["nodes"] >= 4, f"{where}: only {sf['nodes']} fused outputs were visible"
        assert hu.rel_err(ya, yc) <= pp.TAINT_TOL
        for k in ("nodes", "clean", "tainted", "flips", "sites"):
            tot[k] += st[k]
        tot["hidden"] += sf["hidden"]; tot["fused_nodes"] += sf["nodes"]; tot["fused_clean"] += sf["clean"]
        worst_clean = max(worst_clean, st["worst_clean"], sf["worst_clean"])
        worst_tainted = max(worst_tainted, st["worst_tainted"], sf["worst_tainted"])
        flipped_sites.extend((where,) + s for s in st["site_flips"] if s[3])
        return ya

    # Temporal: the oracle's own activations chained through all 32 layers, at ring position 0 (one live slot) and 1 (two: a real soft-max)
    for offset in (0, 1):
        x = (rng.standard_normal(cfg.dim) * 4).astype(np.float32)   # ~ the sum of 17 unit-variance embedding rows
        for layer in range(cfg.num_layers):
            x = one(0, layer, 0, x, offset, f"temporal layer {layer} offset {offset}")
    # Depth: the chain's first six steps as the cached graph runs them - step k uses weight set k and ring slot k, and attends to the rows
    # steps 0..k-1 of this same run left in the ring of 8 (lm.h:505-527)
    for step in range(6):
        x = (rng.standard_normal(cfg.dep_dim) * 2).astype(np.float32)
        for layer in range(cfg.dep_layers):
            x = one(1, layer, step, x, step, f"depth layer {layer} step {step}")
    ref.free(); dev.free()
    print("full-width node parity:", tot, f"worst clean {worst_clean:.2e} worst tainted {worst_tainted:.2e}; flipped sites: {flipped_sites[:12]}")
    assert tot["clean"] >= 0.25 * tot["nodes"], "too few nodes were compared without a flip upstream"
    assert tot["flips"] <= 0.002 * 4096 * tot["sites"] / 8, f"{tot['flips']} rounding flips over {tot['sites']} sites"
    assert tot["hidden"] <= 0.25 * tot["fused_nodes"], f"{tot['hidden']} of {tot['fused_nodes']} fused outputs moved without a flip seen in the per-node run"


def test_contractive_full_config_free_running_greedy_is_bit_exact():
    # the benchmark configuration with residual updates scaled down 16x: every kernel shape / type / byte count of bench.py, free-running
    # (each frame's sampled tokens feed the next through the delay ring), 32 frames from an empty ring
    cfg = lm_only(hu.hot.moshika(L))
    cfg.update_scale = 1.0 / 16
    steps = 32
    rng = np.random.default_rng(21)
    inputs = [rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist() for _ in range(steps)]
    rec = {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        r = []
        for ia in inputs:
            out = m.lm_step(ia)
            r.append((out, m.last_raw(), m.read("text_logits", cfg.text_card).copy(), [m.read(f"dep_logits{k}", cfg.card).copy() for k in range(cfg.dep_q)]))
        rec[kind] = r
        if kind == "hip":
            assert m.stats().graph_replays > 0
        m.free()
    te, de = [], []
    for i, (a, b) in enumerate(zip(rec["oracle"], rec["hip"])):
        assert a[0] == b[0] and a[1] == b[1], f"frame {i}: tokens differ: oracle {a[1]} vs device {b[1]}"
        te.append(hu.rel_err(a[2], b[2]))
        de.append(max(hu.rel_err(x, y) for x, y in zip(a[3], b[3])))
        assert te[-1] < 1e-3, f"frame {i}: text logits rel err {te[-1]:.2e}"
        assert de[-1] < 1e-3, f"frame {i}: Depth logits rel err {de[-1]:.2e}"
    toks = {t for a in rec["oracle"] for t in a[1][1]}
    assert len(toks) > 16, "degenerate run: the sampled audio tokens barely vary"
    print(f"contractive full config, {steps} free-running frames: text logits max {max(te):.2e} median {np.median(te):.2e}; depth max {max(de):.2e}")


def test_mimi_encoder_32_levels_133_frames_codes_exact():
    # configs[2] (moshi-stt, 10 s wav): 125 + 8 frames through the encoder with all 32 RVQ levels; the encoder transformer sees T = 2 per frame,
    # so its offset passes the ring capacity 250 at frame 125 and the mask's wrapped branch (torch.h:211-214, the T = 2 quirk) is exercised
    cfg = hu.hot.stt_like(L)
    cfg.enable_lm = 0
    cfg.enable_mimi_decoder = 0
    assert cfg.mimi_n_q == 32
    rng = np.random.default_rng(31)
    t = np.arange(133 * 1920) / 24000.0
    wave = (0.3 * np.sin(2 * np.pi * 220 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 0.7 * t)) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    wave[125 * 1920:] = 0                                    # the 8 tail frames are silence (tools/moshi-stt.cpp:574-577)
    codes = {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        codes[kind] = [m.mimi_encode(wave[i * 1920:(i + 1) * 1920]) for i in range(133)]
        m.free()
    bad = [(i, a, b) for i, (a, b) in enumerate(zip(codes["oracle"], codes["hip"])) if a != b]
    assert not bad, f"{len(bad)} of 133 frames differ, first: frame {bad[0][0]} oracle {bad[0][1]} device {bad[0][2]}"
    assert len({tuple(c) for c in codes["oracle"]}) > 100, "degenerate input: the codes barely vary"


def test_freed_model_then_a_different_config_never_replays_a_stale_plan():
    # plans are cached by graph address and validated by a hash of everything the planner reads; a model that is freed and replaced by one
    # with another configuration (allocations land on recycled host and device addresses) must plan afresh and still match the oracle
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, hot_util as hu
from ggml_util import Q4_K, Q4_0, Q8_0
L = hu.L
def run(kind, cfgs):
    out = []
    for lt, layers, dq in cfgs:
        cfg = hu.hot.tiny(L, linear_type=lt, embed_type=Q4_0, layers=layers, dep_q=dq)
        cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
        m = hu.Model(kind, cfg, seed=0)
        rng = np.random.default_rng(3)
        for _ in range(4):
            r = m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist())
            out.append((r, m.last_raw()))
        m.free()
    return out
cfgs = [(Q4_K, 2, 3), (Q8_0, 2, 3), (Q4_K, 3, 3), (Q4_K, 2, 4), (Q4_K, 2, 3)]
a, b = run("oracle", cfgs), run("hip", cfgs)
assert a == b, [i for i, (x, y) in enumerate(zip(a, b)) if x != y]
print("OK")
'''
    env = dict(os.environ, MI355X_POISON="1", PYTHONPATH=os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
