"""Shared builders for the tests: the same API objects -> descriptors for the oracle and
for the HIP library."""
from __future__ import annotations

import numpy as np

from attpc_engine_amd import _abi, nuclear_map, workloads
from attpc_engine_amd.detector.luts import build_det_desc, build_layout, species_for


class Inputs:
    """Descriptors of one workload.  ``det`` has the beam pads folded (product), ``det_raw``
    has the unfolded LUT (the oracle applies the beam-pad list itself)."""

    def __init__(self, name: str, ode_substeps: int = 1, **kw):
        det_overrides = {k: kw.pop(k) for k in ("path_step",) if k in kw}  # DetectorParams fields, any workload
        self.pipeline, self.config, self.indices = workloads.WORKLOADS[name](**kw)
        for key, value in det_overrides.items():
            setattr(self.config.det_params, key, value)
        self.kin, self._k1 = self.pipeline.device_desc()
        self.z = self.pipeline.get_proton_numbers()
        self.a = self.pipeline.get_mass_numbers()
        self.n_rows = len(self.z)
        if self.config is not None:
            self.species = species_for(self.z, self.a, self.indices)
            nuclei = [nuclear_map.get_data(z, a) for z, a in self.species]
            self.det, self._k2 = build_det_desc(self.config, nuclei, ode_substeps, fold_beam=True)
            self.det_raw, self._k3 = build_det_desc(self.config, nuclei, ode_substeps, fold_beam=False)
            self.layout = build_layout(self.z, self.a, self.indices, self.species)


def sort_cloud(points: np.ndarray, labels: np.ndarray):
    """Canonical order: by (pad, integer time bucket)."""
    pad = points[:, 0].astype(np.int64)
    tb = np.floor(points[:, 1]).astype(np.int64)
    order = np.lexsort((tb, pad))
    return points[order], labels[order]


def compare_clouds(pts_a, lab_a, pts_b, lab_b, charge_tol: float = 2.0):
    """Both sorted.  Keys, labels and jittered time buckets must agree exactly; charges to
    within ``charge_tol`` electrons (integer truncation of a product that differs in the
    last bits between host libm and device libm, see DESIGN.md "Tolerances")."""
    assert pts_a.shape == pts_b.shape, (pts_a.shape, pts_b.shape)
    np.testing.assert_array_equal(pts_a[:, 0], pts_b[:, 0])
    np.testing.assert_array_equal(np.floor(pts_a[:, 1]), np.floor(pts_b[:, 1]))
    np.testing.assert_array_equal(pts_a[:, 1], pts_b[:, 1])  # jitter is Philox-exact
    np.testing.assert_array_equal(lab_a, lab_b)
    diff = np.abs(pts_a[:, 2] - pts_b[:, 2])
    assert diff.max(initial=0.0) <= charge_tol, diff.max()
    return float(diff.max(initial=0.0))
