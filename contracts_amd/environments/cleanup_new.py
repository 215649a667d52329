"""CleanupEnv — drop-in for the reference's environments/cleanup_new.py:59 `CleanupEnv(MapEnv)`,
stepped by the HIP engine.  Same constructor kwargs (cleanup_new.py:60-74), spaces (:90-169), step /
reset dictionaries (:191-267) and `metrics` keys (:186-188,264-266)."""
import numpy as np

from .. import spaces
from .map_env import GridEnvAdapter

CLEANUP_VIEW_SIZE = 7


def __getattr__(name):
    # CLEANUP_MAP: the layout the engine's tables are built from, read from the library on first use
    if name == "CLEANUP_MAP":
        from .._lib import static_map
        return static_map("cleanup")
    raise AttributeError(name)


class CleanupEnv(GridEnvAdapter):
    KIND = "cleanup"
    GRID_SHAPE = (25, 18)
    N_ACTIONS = (8, 9)  # Discrete(8) without the punishment beam, Discrete(9) with it (cleanup_new.py:90-95)
    N_APPLE_CELLS, POTENTIAL_WASTE_AREA = 103, 119

    def _feature_space(self):
        H, W = self.GRID_SHAPE
        n = self.num_agents
        return spaces.Box(low=np.array([0.0] * (12 + n)),
                          high=np.array([H, W, 4, H, W, 4, H, W, H, W, self.N_APPLE_CELLS + 1,
                                         self.POTENTIAL_WASTE_AREA + 1] + [np.inf] * n))

    def _info_entry(self, eaten, second):
        return {"eaten_apples": eaten, "cleaned_squares": second}

    # ---- inspection helpers (cleanup_new.py:351-394), from the engine's state ----
    @property
    def potential_waste_area(self):
        return self.POTENTIAL_WASTE_AREA

    @property
    def waste_points(self):
        """the waste list in its current order — np.random.shuffle permutes it in place whenever waste may spawn
        (cleanup_new.py:339); the engine keeps the permutation"""
        pts = [[r, c] for r, row in enumerate(self._static_rows) for c, x in enumerate(row) if x in "HR"]
        return [pts[i] for i in self._engine.download("waste_perm")[0][:len(pts)]]  # (the buffer is 119 wide for every layout)

    @property
    def current_waste_points(self):
        return self._cells(b"H")

    def compute_current_wastes(self):
        """the reference rebuilds its `current_waste_points` attribute here (cleanup_new.py:387-394); the property above reads
        the engine's map whenever it is asked, so there is nothing to refresh — kept so that callers of the method still work"""

    def compute_permitted_area(self):
        return self.POTENTIAL_WASTE_AREA - int((self.world_map == b"H").sum())

    def compute_probabilities(self):
        """cleanup_new.py:351-368 on the map as it is now (the reference refreshes these two attributes at the start of
        custom_map_update, i.e. before the step's own spawn)"""
        density = 1 - self.compute_permitted_area() / self.POTENTIAL_WASTE_AREA
        if density >= 0.4:
            self.current_apple_spawn_prob = self.current_waste_spawn_prob = 0
        else:
            self.current_waste_spawn_prob = 0.5
            self.current_apple_spawn_prob = 0.05 if density <= 0.0 else (1 - (density - 0.0) / (0.4 - 0.0)) * 0.05
