"""Episode metrics dictionaries — the engine's metric rows (CE_MI_* / CE_MF_* of include/contracts_engine.h) under the
reference's keys: cleanup_new.py:186-188,264-266, harvest_new.py:152-156,235-237, harvest_features.py:283-298,
cleanup_features.py, self_driving_car_accelerate.py:105 and the wrapper's two_stage_train.py:94-99.  One place for the
single-env adapters and the vector hook (what `MetricsCallback`, utils/logger_utils.py:126-150, logs per episode)."""

GRID_KINDS = ("cleanup", "harvest")


def episode_metrics(kind, n, mi, mf, final, contract=False, inequity=False):
    """kind: engine family; mi / mf: one env's int / float metric rows (the `final_*` rows when `final`: the values
    at the step that ended the episode, with equality / sustainability filled in)"""
    if kind == "selfdrive":
        return {"transfers": float(mf[0])}
    # under inequity aversion the env's rewards are floats and the reference sums those (cleanup_new.py:229-234)
    raw = float(mf[5 + 2 * n]) if inequity else (int(mi[1]) if kind in GRID_KINDS else float(mi[1]))
    transfers = float(mf[0]) if contract else 0
    if kind == "cleanup":
        m = {"total_apples_eaten": int(mi[0]), "raw_env_rewards": raw, "transfers": transfers, "dirt_cleaned": int(mi[2])}
        for i in range(n):
            m["a%d-waste_cleaned" % i] = int(mi[4 + i])
    elif kind == "harvest":
        m = {"total_apples_eaten": int(mi[0]), "low_density_apples_eaten": int(mi[3]), "raw_env_rewards": raw,
             "transfers": transfers}
        for i in range(n):
            m["a%d-apples_consumed" % i] = int(mi[4 + i])
            m["a%d-close_apples_consumed" % i] = int(mi[4 + n + i])
    elif kind == "harvest_features":
        m = {"total_apples_eaten": int(mi[0]), "low_density_apples_eaten": int(mi[3]), "raw_env_rewards": raw,
             "transfers": transfers}
    elif kind == "cleanup_features":
        m = {"dirt_cleaned": int(mi[2]), "raw_env_rewards": raw, "transfers": transfers}
    else:
        raise KeyError(kind)
    if final:
        m["equality"], m["sustainability"] = float(mf[1]), float(mf[2])
        if contract:
            m["transfer_equality"], m["transfer_sustainability"] = float(mf[3]), float(mf[4])
    return m
