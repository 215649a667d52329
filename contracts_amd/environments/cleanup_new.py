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
