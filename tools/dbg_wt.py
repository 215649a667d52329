"""debug: observation bytes of the same rollout with nontemporal vs write-through view stores (CE_OBS_WT_MAX_BYTES read per process)"""
import os
import subprocess
import sys
import numpy as np

if len(sys.argv) > 1:
    sys.path.insert(0, ".")
    import torch
    from contracts_amd.engine import BatchedEnv
    kind, n, Eg, T, seed0 = "cleanup", 8, 16384, 26, 73907
    env = BatchedEnv(kind, Eg, n, contract="cleanup", auto_reset=True, horizon=11)
    acts = torch.empty((T, Eg, n), dtype=torch.uint8, device="cuda")
    env.synth_actions(seed0 + 1, 0, T, acts.data_ptr())
    env.seed(seed0=seed0)
    env.reset()
    streams = [torch.cuda.Stream() for _ in range(3)]
    env.rollout_device(acts.data_ptr(), T, [s.cuda_stream for s in streams])
    torch.cuda.synchronize()
    raw = env.download("obs", raw=True) if "raw" in env.download.__code__.co_varnames else env.download("obs")
    from contracts_amd.engine import _DevArray
    t = torch.as_tensor(_DevArray(env.b.obs, (Eg, env.b.obs_env_stride), np.uint8, None, env), device="cuda").cpu().numpy()
    np.save(sys.argv[1], t)
    sys.exit(0)
for tag, thr in (("nt", "0"), ("wt", "1000000000000")):
    e = dict(os.environ, CE_OBS_WT_MAX_ENVS=thr, CE_OBS_WT_MAX_BYTES=thr)
    subprocess.check_call([sys.executable, __file__, "/tmp/obs_%s.npy" % tag], env=e)
a, b = np.load("/tmp/obs_nt.npy"), np.load("/tmp/obs_wt.npy")
d = np.argwhere(a != b)
print("differing bytes:", len(d))
if len(d):
    print("first", d[:10].tolist())
    off = d[:, 1]
    print("agent", np.unique(off // 720)[:12], "row", np.unique((off % 720) // 48)[:16], "col byte", np.unique(off % 48)[:48])
    print("envs", np.unique(d[:, 0])[:10], len(np.unique(d[:, 0])))
    e0, o0 = d[0]
    print("nt", a[e0, o0 - 4:o0 + 8], "wt", b[e0, o0 - 4:o0 + 8])
