"""RCCL once (review r5 item 6): a world-size-1 `nccl` process group on the GPU box drives GradSync and ShardedGradSync through their REAL
calls -- per-segment async `all_reduce` / `reduce_scatter_tensor` issued with the engine's side ExternalStream current and emit mode on,
`global_sumsq`, `gather_arena`, the ranged fused AdamW, `all_gather_into_tensor` -- and one trainer step through each must leave the same
bits as the step without any exchange.  No byte crosses a link, but it is this code's first contact with RCCL's stream semantics, dtype
and alignment checks (reference: core/distributed.py:88-131, :153-157).  Worker: tests/_rccl_worker.py (its own process: the process
group is per process)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_single_rank_nccl_exchange_bit_equal_to_no_sync():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561",
               HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "_rccl_worker.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "backend nccl world 1 part A ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "RCCL_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
