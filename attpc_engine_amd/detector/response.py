"""GET electronics response (reference ``detector/response.py``).

``get_response`` evaluates the 512-sample closed form once per writer (configure time,
host numpy).  Scaling a point cloud by it -- amplitude and integral with the 4095 clip,
reference response.py:35-57 -- runs on the device in ``attpc_spyral_rows``."""
import numpy as np

from .constants import E_CHARGE, NUM_TB
from .parameters import Config


def get_response(config: Config) -> np.ndarray:
    """Response per electron at ``linspace(0, 512, 512)`` time buckets, negative lobes
    clamped to zero (reference response.py:8-32)."""
    elec = config.elec_params
    scale = 4095 * E_CHARGE / elec.amp_gain / 1e-15
    tau = np.linspace(0.0, NUM_TB, NUM_TB) / (elec.shaping_time * elec.clock_freq * 0.001)
    response = scale * np.exp(-3.0 * tau) * (tau**3) * np.sin(tau)
    response[response < 0] = 0
    return response
