"""HarvestEnv — drop-in for the reference's environments/harvest_new.py:48 `HarvestEnv(MapEnv)`,
stepped by the HIP engine.  Same constructor kwargs (harvest_new.py:49-63), spaces (:85-130), step /
reset dictionaries (:158-239) and `metrics` keys (:152-156,235-237)."""
import numpy as np

from .. import spaces
from .map_env import GridEnvAdapter

HARVEST_VIEW_SIZE = 7


def __getattr__(name):
    # HARVEST_MAP: the layout the engine's tables are built from, read from the library on first use
    if name == "HARVEST_MAP":
        from .._lib import static_map
        return static_map("harvest")
    raise AttributeError(name)


class HarvestEnv(GridEnvAdapter):
    KIND = "harvest"
    GRID_SHAPE = (16, 38)
    N_ACTIONS = (7, 8)  # harvest_new.py:85-90
    N_APPLE_CELLS = 155

    def _feature_space(self):
        H, W = self.GRID_SHAPE
        n = self.num_agents
        return spaces.Box(low=np.array([0.0] * (10 + 2 * n)),
                          high=np.array([H, W, 4, H, W, 4, H, W, self.N_APPLE_CELLS + 1, self.N_APPLE_CELLS + 1]
                                        + [1] * (2 * n)))

    def _info_entry(self, eaten, second):
        return {"eaten_apples": eaten, "eaten_close_apples": second}

    def count_apples_in_radius(self, radius, loc):
        """harvest_new.py: apples with j^2 + k^2 <= radius around loc (the env's own feature uses radius 5)"""
        grid = self.world_map
        H, W = self.GRID_SHAPE
        total = 0
        for j in range(-radius, radius + 1):
            for k in range(-radius, radius + 1):
                r, c = loc[0] + j, loc[1] + k
                if j ** 2 + k ** 2 <= radius and 0 <= r < H and 0 <= c < W and grid[r, c] == b"A":
                    total += 1
        return total
