# coding: utf-8
"""The ragged epoch replayed from a lattice of padded shapes (bench.ragged_lattice_config) beside the eager ragged
epoch (bench.ragged_epoch_config), batch 64 and 16, one process.  -> one JSON object per line.
usage: python scripts/r6_lattice_probe.py [n_batches_64] [n_batches_16] [text_step] [decoder_step]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    n64 = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    n16 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    lat = (int(sys.argv[3]) if len(sys.argv) > 3 else 32, int(sys.argv[4]) if len(sys.argv) > 4 else 16)
    dev = torch.device("cuda:0")
    args = argparse.Namespace(batch=64, text_len=150, frames=800, preset="deepvoice3_ljspeech")
    gemm = "f16x3"
    for batch, nb in ((64, n64), (16, n16)):
        if nb <= 0:
            continue
        args.batch = batch
        forms = (("eager_batch_maxima", lambda: bench.ragged_epoch_config(dev, args.preset, gemm, args)),) \
            if os.environ.get("LATTICE_ONLY", "") in ("", "0") else ()
        for name, fn in forms + (("lattice_replay", lambda: bench.ragged_lattice_config(dev, args.preset, gemm, args, batch,
                                                                                n_batches=nb, lattice=lat)),):
            try:
                out = fn()
            except Exception as e:
                import traceback
                traceback.print_exc()
                out = dict(error="%s: %s" % (type(e).__name__, e))
            out.pop("lengths", None)
            print(json.dumps(dict(batch=batch, form=name, **out)), flush=True)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        print(json.dumps(dict(batch=batch, max_memory_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))), flush=True)


if __name__ == "__main__":
    main()
