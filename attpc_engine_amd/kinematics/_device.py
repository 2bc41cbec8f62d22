"""Host-side marshalling of a kinematics pipeline into the C-ABI descriptor."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _abi

ELOSS_NODES = 2049


def _fill_excitation(dst: _abi.ExcitationDesc, spec: dict, keep: list) -> None:
    dst.kind = int(spec["kind"])
    dst.p0 = float(spec.get("p0", 0.0))
    dst.p1 = float(spec.get("p1", 0.0))
    dst.p2 = float(spec.get("p2", 0.0))
    dst.table_len = 0
    if dst.kind == _abi.EX_TABLE:
        x = np.ascontiguousarray(spec["table_x"], dtype=np.float64)
        cdf = np.ascontiguousarray(spec["table_cdf"], dtype=np.float64)
        if x.shape != cdf.shape or x.size < 2:
            raise ValueError("excitation table needs matching x/cdf arrays of length >= 2")
        keep += [x, cdf]
        dst.table_len = x.size
        dst.table_x = _abi.dptr(x)
        dst.table_cdf = _abi.dptr(cdf)


def _fill_polar(dst: _abi.PolarDesc, spec: dict, keep: list) -> None:
    dst.kind = int(spec["kind"])
    dst.cos_min = float(spec.get("cos_min", 0.0))
    dst.cos_max = float(spec.get("cos_max", 0.0))
    dst.bin_width = float(spec.get("bin_width", 0.0))
    dst.table_len = 0
    if dst.kind == _abi.POLAR_ARBITRARY:
        angles = np.ascontiguousarray(spec["angles"], dtype=np.float64)
        cdf = np.ascontiguousarray(spec["cdf"], dtype=np.float64)
        if angles.shape != cdf.shape or angles.size < 1:
            raise ValueError("PolarArbitrary needs matching angles/probabilities arrays")
        keep += [angles, cdf]
        dst.table_len = angles.size
        dst.angles = _abi.dptr(angles)
        dst.cdf = _abi.dptr(cdf)


def masses_in_row_order(reaction, decays) -> list[float]:
    """target, projectile, ejectile, residual, then (residual_1, residual_2) per decay --
    the row order of the result array (reference pipeline.py:398-406)."""
    masses = [
        reaction.target.mass, reaction.projectile.mass, reaction.ejectile.mass,
        reaction.residual.mass,
    ]
    for decay in decays:
        masses += [decay.residual_1.mass, decay.residual_2.mass]
    return [float(m) for m in masses]


def is_device_samplable(pipeline) -> bool:
    return all(hasattr(d, "device_desc") for d in pipeline.excitations) and all(
        hasattr(d, "device_desc") for d in pipeline.polar_dists
    )


def build_kin_desc(pipeline, deterministic_only: bool = False):
    """-> (KinDesc, keepalive list).  ``deterministic_only`` fills just masses/n_steps (the
    part ``attpc_kin_calculate`` needs) for pipelines with custom Python distributions."""
    n_steps = 1 + len(pipeline.decays)
    if n_steps > _abi.MAX_STEPS:
        raise ValueError(f"at most {_abi.MAX_STEPS} steps are supported on the device")
    keep: list = []
    desc = _abi.KinDesc()
    desc.n_steps = n_steps
    desc.sample_limit = int(pipeline.event_sample_limit)
    desc.beam_energy = float(pipeline.beam_energy)
    for i, m in enumerate(masses_in_row_order(pipeline.reaction, pipeline.decays)):
        desc.masses[i] = m
    desc.has_target = 0
    desc.eloss_len = 0
    if deterministic_only:
        for s in range(n_steps):
            desc.excitation[s].kind = _abi.EX_UNIFORM
            desc.polar[s].kind = _abi.POLAR_UNIFORM
        return desc, keep
    for s in range(n_steps):
        _fill_excitation(desc.excitation[s], pipeline.excitations[s].device_desc(), keep)
        _fill_polar(desc.polar[s], pipeline.polar_dists[s].device_desc(), keep)
    tm = pipeline.target_material
    if tm is not None:
        z0, z1 = float(tm.z_range[0]), float(tm.z_range[1])
        grid = np.linspace(z0, z1, ELOSS_NODES)
        # one energy-loss integration per node at configure time replaces one per sample
        # (reference pipeline.py:256-264 calls get_energy_loss for every attempt)
        eloss = np.ascontiguousarray(
            np.asarray(
                tm.material.get_energy_loss(pipeline.reaction.projectile, pipeline.beam_energy, grid),
                dtype=np.float64,
            ).reshape(-1)
        )
        keep.append(eloss)
        desc.has_target = 1
        desc.eloss_len = eloss.size
        desc.rho_sigma = float(tm.rho_sigma)
        desc.z_min = z0
        desc.z_max = z1
        desc.eloss = _abi.dptr(eloss)
    return desc, keep


def decay_only_calculate(decay, parent_vector, polar, azim, excitation, ctx=None):
    """One Decay.calculate on the device -> (rows [2,4], allowed)."""
    ctx = ctx or _abi.default_context()
    parent = np.array(
        [[parent_vector.px, parent_vector.py, parent_vector.pz, parent_vector.E]], dtype=np.float64
    )
    ex = np.array([excitation], dtype=np.float64)
    th = np.array([polar], dtype=np.float64)
    ph = np.array([azim], dtype=np.float64)
    out = np.empty((1, 2, 4), dtype=np.float64)
    status = np.empty(1, dtype=np.int32)
    ctx.check(
        ctx.lib.attpc_decay_calculate(
            ctx.handle, 1, _abi.dptr(parent), float(decay.residual_1.mass),
            float(decay.residual_2.mass), _abi.dptr(ex), _abi.dptr(th), _abi.dptr(ph),
            _abi.dptr(out), _abi.iptr(status, C.c_int32),
        ),
        "attpc_decay_calculate",
    )
    return out[0], bool(status[0] == 0)
