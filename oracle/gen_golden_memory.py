"""Generates tests/golden/memory_bank.npz by driving the REAL reference methods (StreamPETRHead.pre/post_update_memory)
through oracle/ref_harness.ReferenceMemory.  Runs only in the build container (needs /root/reference).
Sequence: frame 0 scene start (prev_exists = 0), frames 1-2 continuation, frame 3 scene change for batch element 1 only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as R          # noqa: E402
from toc3d_amd import synth                  # noqa: E402

CFG = dict(memory_len=48, topk_proposals=16, num_propagated=16, embed_dims=32, pc_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0])
B, NQ, NCLS, FRAMES = 2, 70, 10, 4


def main():
    out = {}
    inp = synth.memory_inputs(CFG, B, NQ, NCLS, FRAMES, seed=0)
    m = R.ReferenceMemory(pseudo_reference_points=inp["pseudo"], **CFG)
    for f in range(FRAMES):
        fr = inp["frames"][f]
        m.pre_update_memory(fr["data"])
        for k in ("embedding", "reference_point", "timestamp", "egopose", "velo"):
            out[f"f{f}_pre_{k}"] = getattr(m, "memory_" + k).numpy().copy()
        m.post_update_memory(fr["data"], fr["rec_ego_pose"], fr["cls"][None], fr["bbox"][None], fr["dec"][None])
        for k in ("embedding", "reference_point", "timestamp", "egopose", "velo"):
            out[f"f{f}_post_{k}"] = getattr(m, "memory_" + k).numpy().copy()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "memory_bank.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.dtype for k, v in list(out.items())[:12]})


if __name__ == "__main__":
    main()
