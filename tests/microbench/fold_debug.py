"""debug: folded attention vs two launches vs oracle, one Temporal layer, first steps"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hot_util as hu
import test_attn_fold as tf

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 24
fill = int(sys.argv[2]) if len(sys.argv) > 2 else None
cfg = tf.temporal_cfg(ctx, layers=1)
fold, st = tf.run("hip", cfg, 3, fill=fill)
plain, _ = tf.run("hip", cfg, 3, flags=256, fill=fill)
ref, _ = tf.run("oracle", cfg, 3, fill=fill)
print("folds planned", st.attention_folds_planned)
for i in range(3):
    f, p, r = fold[i][4], plain[i][4], ref[i][4]
    print(f"step {i}: fold-vs-plain max {np.abs(f - p).max():.3e}  plain-vs-oracle {np.abs(p - r).max():.3e}  fold-vs-oracle {np.abs(f - r).max():.3e}")
    bad = np.nonzero(f != p)[0]
    print("   differing rows:", len(bad), bad[:40])
