"""Rehearsal of the sharded pipeline with two ranks sharing cuda:0 (gloo transports the
collectives; RCCL refuses two ranks on one device).  Checks rank results against the oracle."""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from bench import synth, MATCH_THR_ABS, MATCH_RATIO, RANSAC_COEF
from pcreg_amd.device import RegistrationPipeline, soa
Q, M_local = 6000, 40000
model, surf, _ = synth(M_local * world, Q)
dev = torch.device("cuda", 0)
m_lo = rank * M_local
ms = soa(torch.from_numpy(model[m_lo:m_lo + M_local]).to(dev)); qs = soa(torch.from_numpy(surf).to(dev))
pipe = RegistrationPipeline(Q, M_local, m_lo=m_lo, M_total=M_local * world, device=dev)
pairs, p1, p2, n = pipe.match(qs, ms, MATCH_THR_ABS, MATCH_RATIO, True)
pipe.ransac_sharded(dict(RANSAC_COEF, iterNum=1001), seed=7)      # hypotheses split over the two ranks
res = pipe.fetch_result(); n = int(n.item())
from oracle import c_oracle
ref = c_oracle.match_points_f32(surf, model, MATCH_THR_ABS, MATCH_RATIO, True)
ok = np.array_equal(pairs[:n].cpu().numpy().astype(np.uint32), ref)
rp1 = surf[ref[:, 0] - 1].astype(np.float64); rp2 = model[ref[:, 1] - 1].astype(np.float64)
rr = c_oracle.ransac(rp1, rp2, dict(RANSAC_COEF, iterNum=1001), seed=7)
ok2 = (np.array_equal(res["inlierIdx"].astype(np.int64), rr["inlierIdx"]) and res["numSuccess"] == rr["numSuccess"]
       and res["maxInliers"] == rr["maxInliers"] and res["winner"] == int(np.argmax(rr["inlrNum_refined"])) and np.linalg.norm(res["T"] - rr["T"]) < 1e-9)
# two registrations in flight per rank (pcreg_amd/pipelined.py): four different surfaces dealt to two lanes, each lane issuing its
# three collectives in host program order -- every result must be the serial pipeline's for that surface, bit for bit
from pcreg_amd.pipelined import PipelinedRegistration
coef = dict(RANSAC_COEF, iterNum=1001)
surfs = [qs] + [soa(torch.from_numpy(surf + np.float32(0.01 * k) * np.array([1, -1, 0.5], np.float32)).to(dev)) for k in range(1, 4)]
serial = []
for q in surfs:
    pipe.match(q, ms, MATCH_THR_ABS, MATCH_RATIO, True); pipe.ransac_sharded(coef, seed=7)
    r = pipe.fetch_result(); r["n_pairs"] = int(pipe.n_pairs.item()); serial.append(r)
pr = PipelinedRegistration(Q, M_local, lanes=2, m_lo=m_lo, M_total=M_local * world, device=dev)
piped = []
for k in (0, 2):                                   # two submissions, then their results (a lane keeps only its last result)
    pr.submit(surfs[k], ms, MATCH_THR_ABS, MATCH_RATIO, coef, seed=7); pr.submit(surfs[k + 1], ms, MATCH_THR_ABS, MATCH_RATIO, coef, seed=7)
    piped += pr.results()
ok3 = len(piped) == 4
for a, b in zip(serial, piped):
    ok3 = ok3 and a["n_pairs"] == b["n_pairs"] and a["numSuccess"] == b["numSuccess"] and a["maxInliers"] == b["maxInliers"] and a["winner"] == b["winner"]
    ok3 = ok3 and np.array_equal(a["inlierIdx"], b["inlierIdx"]) and np.array_equal(a["T"], b["T"])
print(f"rank {rank}: pairs {n} match_ok={ok} ransac_ok={ok2} pipelined_ok={ok3}", flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok and ok2 and ok3 else 1)
