#!/usr/bin/env python3
"""Write tests/golden/sdxl_segments.json: the backward segments (arena offset, element count; exchange order) and the arena size of the
SDXL-base UNet as the native engine lays them out.  Needs a GPU (sdxl_create); tests/test_gpu_model.py checks the live values against the
file, tests/test_distributed_cpu.py drives the world-8 slicing arithmetic over them.   python profiles/tools/dump_segments.py"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import unet as NU  # noqa: E402

net = NU.NativeUNet(NU.make_config(), 0)
out = {"param_elems": int(net.param_elems), "segments": [[int(o), int(n)] for o, n in net.segment_ranges()]}
(ROOT / "tests" / "golden" / "sdxl_segments.json").write_text(json.dumps(out))
if (ROOT / "gpurun_out").is_dir():                     # (a gpurun box only ships gpurun_out/ back)
    (ROOT / "gpurun_out" / "sdxl_segments.json").write_text(json.dumps(out))
print(out["param_elems"], len(out["segments"]), sum(n for _o, n in out["segments"]))
net.close()
