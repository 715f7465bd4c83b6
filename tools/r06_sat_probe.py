#!/usr/bin/env python
"""Round 6: batches per launch group -> Mpixels/s (bench.saturation_curve) with the pipelined decoder's chain on the full
image, on the compact image and chosen by launch (tfc_set_pipe_format); optional waves per workgroup.
Usage: python tools/r06_sat_probe.py [points, default 20,32,64,128] [formats, default 0,1,2] [waves, default 0]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from compression_amd import _lib  # noqa: E402

points = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "20,32,64,128").split(",")]
formats = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2").split(",")]
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
lookup = bench.build_tables(device)
lookup_t = torch.from_numpy(lookup)
for fmt in formats:
    _lib.lib().tfc_set_pipe_format(fmt, waves)
    res = bench.saturation_curve(lookup, lookup_t, device, points, bytes_per_batch=12_600_000)
    for row in res["points"]:
        print(json.dumps({"format": fmt, "waves": waves, **{k: row[k] for k in (
            "batches_per_launch", "mpixels_s", "ms_per_group", "encode_call_ms", "decode_call_ms", "kernels_ms",
            "launches_per_direction")}}))
    torch.cuda.empty_cache()
