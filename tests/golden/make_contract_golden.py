"""Golden vectors for `Contract.compute_transfer` of the three shipped contracts (contract/contract_list.py:22-27,
45-54,69-102): random one-step inputs -> the reference's transfer dictionaries, flattened to arrays.  Runs ONLY in the
build container (imports the reference through ref_harness); commits tests/golden/contract_transfers.npz (data only).

    python tests/golden/make_contract_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402


def flatten(tr, n_slots):
    """{agent: number | (value, {recipient: share})} -> value[n_slots], is_tuple[n_slots], share[n_slots, n_slots]"""
    val, tup, share = np.zeros(n_slots), np.zeros(n_slots, np.uint8), np.zeros((n_slots, n_slots))
    present = np.zeros(n_slots, np.uint8)
    for k, v in tr.items():
        i = int(k[1:])
        present[i] = 1
        if type(v) is tuple:
            tup[i] = 1
            val[i] = v[0]
            for r, p in v[1].items():
                share[i, int(r[1:])] = p
        else:
            val[i] = v
    return val, tup, share, present


def main():
    ref = ref_harness.load_reference()
    cl = ref.contract_list
    rs = np.random.RandomState(20260)
    out = {}
    # grid contracts: n agents, all acting
    for name, cls in (("cleanup", cl.CleanupContract), ("harvest", cl.HarvestFeaturemodLocalContract)):
        cases = []
        for n in (2, 4, 8):
            c = cls(n)
            for _ in range(40):
                keys = ["a%d" % i for i in range(n)]
                theta = rs.uniform(0, 0.2 if name == "cleanup" else 10.0)
                params = {k: np.array([theta]) for k in keys}
                cleaned = rs.randint(0, 4, size=n)
                f8 = rs.randint(0, 9, size=n)
                close = rs.randint(0, 2, size=n)
                infos = {k: {"cleaned_squares": int(cleaned[i]), "eaten_close_apples": int(close[i]),
                             "feature_obs": np.r_[np.zeros(8), f8[i], np.zeros(3)]} for i, k in enumerate(keys)}
                acts = {k: 0 for k in keys}
                tr = c.compute_transfer({}, acts, {k: 0 for k in keys}, params, infos)
                val, tup, share, present = flatten(tr, 8)
                cases.append(np.r_[n, theta, np.pad(cleaned, (0, 8 - n)), np.pad(f8, (0, 8 - n)), np.pad(close, (0, 8 - n)), val,
                                   present])
        out[name] = np.array(cases)
    # selfdrive: subsets of acting agents, ambulance passing or not, cars ahead / behind
    cases = []
    for n in (2, 4, 6):
        c = cl.SelfdriveContractDistprop(n)
        S = n + 2  # slots the reference's loop bound len(obs) // 2 reaches
        for _ in range(60):
            keys = ["a%d" % i for i in range(n)]
            theta = rs.uniform(0, 100.0)
            rel = np.r_[0.0, rs.uniform(-3, 3, size=n - 1)]
            if rs.rand() < 0.3:
                rel = np.abs(rel)  # nobody behind
            vel = rs.uniform(0, 1, size=n)
            row = np.r_[rs.uniform(-1, 1), rs.uniform(0, 1), rel, vel, 1.0, 1.0, 0.0]  # [p, v, rel.., vel.., flags, 0]: 2n + 5
            acting = rs.rand(n) < 0.8
            acting[0] = rs.rand() < 0.85
            passed = bool(rs.rand() < 0.7)
            acts = {k: 0 for i, k in enumerate(keys) if acting[i]}
            if not acts:
                acts = {"a1": 0}
                acting[1] = True
            obs = {k: row.copy() for k in acts}
            obs["a0"] = row.copy()
            infos = {k: {"just_passed": passed if k == "a0" else False} for k in keys}
            params = {k: np.array([theta]) for k in keys}
            tr = c.compute_transfer(obs, acts, {}, params, infos)
            val, tup, share, present = flatten(tr, 12)
            cases.append(np.r_[n, theta, passed, np.pad(acting.astype(float), (0, 8 - n)), np.pad(row, (0, 20 - len(row))), val, tup,
                               share.reshape(-1), present])
    out["selfdrive"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "contract_transfers.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
