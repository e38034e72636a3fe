"""World size 2 with the REAL library on ONE GPU.  RCCL refuses two ranks on one device, so the two
processes -- both on cuda:0 -- exchange over a gloo group with the packed head / tail staged through pinned host memory
(``cca_zoo_amd._dist.staged_over_gloo``); the two-part order, the deferred tail behind ``ccz_solve_defer`` and every
rank-dependent line (ragged ``shard_bounds``, ``row0`` into the device generator, sharded ``score``, the sharded loss,
rank-0-only comparisons) run exactly as on the nccl route.  Rank 0 compares rCCA / MCCA / GCCA / CCALoss from two ragged
shards with the single-process results at 1e-10.

Reference seams: cca_zoo/linear/_rcca.py:69-101, _mcca.py:99-197, _gcca.py:80-110 on row shards; SURVEY.md 8(e)."""

import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RANK = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch
import torch.distributed as dist
rank, world = int(sys.argv[1]), int(sys.argv[2])
torch.cuda.set_device(0)                                    # BOTH ranks on the one GPU
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
from cca_zoo_amd import _backend, _dist, row_sharded, shard_bounds
from cca_zoo_amd.datasets import JointData
from cca_zoo_amd.deep.objectives import CCALoss
from cca_zoo_amd.linear import GCCA, MCCA, rCCA
ok = True
def check(name, err, tol=1e-10):
    global ok
    ok = ok and (err < tol)
    print(name, err, flush=True)

n = 6001                                                    # ragged shards: 3001 + 3000 rows
jd = JointData(n_views=3, n_samples=n, latent_dimensions=4, n_features=[300, 260, 200], signal_to_noise=2.0, random_state=21,
               latent_scales=[2.0, 1.5, 1.0, 0.5])
lo, hi = shard_bounds(n, rank, world)
# the device generator at this rank's row offset: the same global data set whatever the number of ranks
mine64 = jd.sample_device(device="cuda:0", dtype=torch.float64, n_samples=hi - lo, seed=5, row0=lo)
full64 = jd.sample_device(device="cuda:0", dtype=torch.float64, n_samples=n, seed=5, row0=0) if rank == 0 else None
if rank == 0:
    for a, b in zip(mine64, full64):
        check("generator rows at row0", float((a - b[lo:hi]).abs().max()), 1e-300)
for name, make, nv in (("rcca", lambda: rCCA(latent_dimensions=4, c=[0.1, 0.2]), 2),
                       ("mcca", lambda: MCCA(latent_dimensions=4, c=0.1), 3),
                       ("gcca", lambda: GCCA(latent_dimensions=4, c=0.05, view_weights=[1.0, 2.0, 0.5]), 3)):
    for on_device in (True, False):
        mine = [v if on_device else v.cpu().numpy() for v in mine64[:nv]]
        with row_sharded():
            m = make().fit(mine)                            # defer_offdiag: the tail is unpacked behind ccz_solve_defer
            sc = m.score(mine)
        assert m.n_samples_ == n
        if rank == 0:
            ref = make().fit([v for v in full64[:nv]])
            for a, b in zip(ref.weights_, m.weights_):
                check(f"{{name}} {{'device' if on_device else 'host'}} weights", float(np.abs(np.asarray(b) - a).max() / np.abs(a).max()))
            check(f"{{name}} score", float(np.abs(np.asarray(sc) - ref.score([v for v in full64[:nv]])).max()))
# the sharded two-view loss: moments all-reduced, gradients of THIS rank's rows
z = [v[:, :96].clone().requires_grad_(True) for v in mine64[:2]]
with row_sharded():
    loss = CCALoss(eps=1e-4)(z)
    loss.backward()
if rank == 0:
    zf = [v[:, :96].clone().requires_grad_(True) for v in full64[:2]]
    lf = CCALoss(eps=1e-4)(zf)
    lf.backward()
    check("loss", abs(float(loss) - float(lf)) / abs(float(lf)))
    for a, b in zip(z, zf):
        check("loss grad rows", float((a.grad - b.grad[lo:hi]).abs().max() / b.grad.abs().max()))
# an error between the exchange and the solve must not leave the deferred tail pending (ADVICE r5)
try:
    with row_sharded():
        rCCA(latent_dimensions=4, c=[0.1, 0.2, 0.3]).fit(mine64[:2])
    ok = False
except ValueError:
    pass
with row_sharded():
    again = rCCA(latent_dimensions=4, c=[0.1, 0.2]).fit(mine64[:2])
if rank == 0:
    ref = rCCA(latent_dimensions=4, c=[0.1, 0.2]).fit(full64[:2])
    check("fit after a failed fit", float(max(np.abs(np.asarray(b) - a).max() / np.abs(a).max() for a, b in zip(ref.weights_, again.weights_))))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok" if ok else "FAILED", flush=True)
sys.exit(0 if ok else 1)
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_on_one_gpu_match_the_single_process_results():
    world = 2
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "rank.py")
        with open(script, "w") as f:
            f.write(_RANK.format(root=ROOT))
        port = str(_free_port())
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script, str(r), str(world), port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                  text=True, env=env) for r in range(world)]
        outs = [p.communicate(timeout=600)[0] for p in procs]
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and f"rank {r} ok" in o, f"rank {r}:\n{o[-4000:]}"


def test_bench_two_ranks_on_one_gpu_prints_one_json_line():
    """``bench.py --gpus 2 --transport gloo-staged``: the benchmark's own N > 1 path (rank-0-only gates, max-over-ranks timing,
    ONE JSON line) with two ranks on the one GPU; a reduced n keeps it to seconds."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "gloo-staged", "--steps", "2", "--warmup", "1",
           "--no-extras", "--no-cpu-baseline", "--rows", "131072", "--dim", "1024", "--k", "16"]     # (--dim: torch.distributed.run would take "--d" for one of its own options)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]        # ONE line on stdout, and it is the JSON
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["transport"] == "gloo-staged" and rec["ranks_on_one_gpu"] == 2
    assert rec["parity_gate"]["ok"] and rec["value"] > 0
