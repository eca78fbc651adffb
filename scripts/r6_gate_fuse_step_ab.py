# coding: utf-8
"""Round 6: whole steps, replayed, alternating in one process:
  base   the round-5 backward: stand-alone dv3_gate_bwd_f32 launches writing an fp32 pre-gate gradient
  pair   ... writing PAIR WORDS, which the layer's two gradient GEMMs stage without conversion (ops.pair_words)
  rule   pair + the gate backward of small layers run by their consumers' input-gradient launches (ops.GateFuse up to
         ops.fuse_gate_max_elems elements: the default)
  all    pair + every eligible gate backward fused, whatever its size
argv: [preset names ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deepvoice3_pytorch_amd import ops
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cases = (("deepvoice3_ljspeech", "f16x3", 64), ("deepvoice3_ljspeech", "f16x3", 16), ("nyanko_ljspeech", "f16x3", 64),
         ("nyanko_ljspeech", "f16x3", 16))
rounds = int(os.environ.get("AB_ROUNDS", "3"))
RULE = ops.fuse_gate_max_elems
VARIANTS = (("base", False, False, RULE), ("pair", True, False, RULE), ("rule", True, True, RULE), ("all", True, True, 1 << 40))
for preset, gemm, B in cases:
    if len(sys.argv) > 1 and preset not in sys.argv[1:]:
        continue
    res, stats = {}, {}
    for rnd in range(rounds):
        for name, pw, fuse, mx in VARIANTS:
            ops.pair_words, ops.fuse_gate_bwd, ops.fuse_gate_max_elems = pw, fuse, mx
            before = dict(ops.gate_fuse_stats)
            run = bench.TrainRun(dev, None, 0, 1, preset, gemm, B, 150, 800, graph=True)
            m = run.measure(15, 5, settle_s=0.5)
            run.close()
            res.setdefault(name, []).append(round(m["ms_per_step"], 3))
            stats[name] = {k: ops.gate_fuse_stats[k] - before[k] for k in before}
    ops.pair_words, ops.fuse_gate_bwd, ops.fuse_gate_max_elems = True, True, RULE
    print(preset, gemm, "B=%d" % B, " ".join("%s %s" % (n, res[n]) for n, _, _, _ in VARIANTS),
          "| gated backwards while capturing (rule):", stats["rule"], "(all):", stats["all"], flush=True)
