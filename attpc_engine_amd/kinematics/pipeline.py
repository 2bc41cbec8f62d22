"""Kinematics pipeline (reference ``kinematics/pipeline.py``), device backed.

The reference samples one event per ``run()`` call in a serial Python loop.  Here a
pipeline is marshalled once into a C descriptor, events are generated on the GPU in
batches (one lane per event, counter-based Philox streams keyed by the *global event
id*), and ``run()`` hands them out one at a time, so scripts written against the
reference keep working while ``run_many`` / ``run_kinematics_pipeline`` get the batch
rate.  Unlike the reference (unseeded ``default_rng()``, pipeline.py:152) a ``seed`` can
be given; the default draws fresh entropy.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path

import numpy as np
from numpy.random import default_rng

from .. import _abi
from ._device import build_kin_desc, is_device_samplable
from .angle import PolarDistribution
from .excitation import ExcitationDistribution
from .reaction import Decay, Reaction, device_calculate

CHUNK_SIZE: int = 1_000_000  # events per HDF5 chunk group, reference pipeline.py:13


@dataclass
class KinematicsTargetMaterial:
    """Target gas + vertex sampling ranges (reference pipeline.py:16-36).

    ``material`` is any object with ``get_energy_loss(projectile, energy, distances_m)``;
    ``z_range`` (m) bounds the uniformly sampled vertex z and the beam path length;
    ``rho_sigma`` (m) is the sigma of the half-normal radial vertex distribution."""

    material: object
    z_range: tuple[float, float]
    rho_sigma: float


@dataclass
class Sample:
    """One set of sampled pipeline parameters (reference pipeline.py:39-70)."""

    beam_energy: float
    reaction_excitation: float
    reaction_theta: float
    reaction_phi: float
    vertex: np.ndarray
    decay_excitations: list[float]
    decay_thetas: list[float]
    decay_phis: list[float]


class PipelineError(Exception):
    """Pipeline construction / sampling error (reference pipeline.py:73-76)."""


class KinematicsPipeline:
    """A Reaction followed by Decays with one excitation and one polar distribution per step.

    Same constructor and validation as the reference (pipeline.py:125-185); extra keyword
    arguments ``seed``, ``batch_size`` and ``context`` are additive.
    """

    def __init__(
        self,
        steps: list[Reaction | Decay],
        excitations: list[ExcitationDistribution],
        polar_dists: list[PolarDistribution],
        beam_energy: float,
        target_material: KinematicsTargetMaterial | None = None,
        event_sample_limit: int = 1000,
        seed: int | None = None,
        batch_size: int = 8192,
        context: _abi.Context | None = None,
    ):
        if len(steps) == 0:
            raise PipelineError("Pipeline must have at least one step (a Reaction)!")
        if len(steps) != len(excitations):
            raise PipelineError(
                f"Pipeline must have the same number of steps (given {len(steps)}) and "
                f"excitations (given {len(excitations)}!"
            )
        if len(steps) != len(polar_dists):
            raise PipelineError(
                f"Pipeline must have the same number of steps (given {len(steps)}) and polar "
                f"angle distributions (given {len(polar_dists)})!"
            )
        if not isinstance(steps[0], Reaction):
            raise PipelineError("The first element in the pipeline must be a Reaction!")

        self.reaction: Reaction = steps[0]
        self.decays: list[Decay] = []
        self.excitations = excitations
        self.polar_dists = polar_dists
        self.event_sample_limit = event_sample_limit

        expected_parent = self.reaction.residual
        for idx in range(1, len(steps)):
            step = steps[idx]
            if not isinstance(step, Decay):
                raise PipelineError(
                    "All elements in the pipeline after the first element must be Decay!"
                )
            if expected_parent.isotopic_symbol != step.parent.isotopic_symbol:
                which = "residual" if idx == 1 else "residual_2"
                raise PipelineError(
                    f"Broken step in pipeline! Step {idx - 1} {which} does not match Step {idx} parent!"
                )
            self.decays.append(step)
            expected_parent = step.residual_2

        returned_nuclei = 4 + (len(steps) - 1) * 2
        self.result = np.empty((returned_nuclei, 4), dtype=float)
        self.beam_energy = beam_energy
        self.target_material = target_material

        # --- device state ---
        self.seed = int(seed) if seed is not None else int(np.random.SeedSequence().entropy % (1 << 63))
        self.rng = default_rng(self.seed)
        self.batch_size = int(batch_size)
        self._ctx = context
        self._configured = False
        self._next_event = 0  # global event id of the next event run() returns
        self._buf_first = 0
        self._buf_vertex: np.ndarray | None = None
        self._buf_p4: np.ndarray | None = None

    # -------------------------------------------------------------- description -----
    def __str__(self) -> str:
        chain = f"{self.reaction}"
        for decay in self.decays:
            chain += f", {str(decay)}"
        return chain

    def get_proton_numbers(self) -> np.ndarray:
        """Z per result row (reference pipeline.py:390-407)."""
        z = [self.reaction.target.Z, self.reaction.projectile.Z, self.reaction.ejectile.Z,
             self.reaction.residual.Z]
        for decay in self.decays:
            z += [decay.residual_1.Z, decay.residual_2.Z]
        return np.array(z, dtype=int)

    def get_mass_numbers(self) -> np.ndarray:
        """A per result row (reference pipeline.py:409-426)."""
        a = [self.reaction.target.A, self.reaction.projectile.A, self.reaction.ejectile.A,
             self.reaction.residual.A]
        for decay in self.decays:
            a += [decay.residual_1.A, decay.residual_2.A]
        return np.array(a, dtype=int)

    def get_nuclei(self) -> list:
        nuclei = [self.reaction.target, self.reaction.projectile, self.reaction.ejectile,
                  self.reaction.residual]
        for decay in self.decays:
            nuclei += [decay.residual_1, decay.residual_2]
        return nuclei

    # ------------------------------------------------------------------ device ------
    @property
    def context(self) -> _abi.Context:
        if self._ctx is None:
            self._ctx = _abi.default_context()
        return self._ctx

    def device_desc(self):
        """(KinDesc, keepalive) -- also what the parity tests hand to the CPU oracle."""
        return build_kin_desc(self)

    def configure_device(self) -> None:
        ctx = self.context
        desc, keep = self.device_desc()
        ctx.check(ctx.lib.attpc_kin_configure(ctx.handle, desc), "attpc_kin_configure")
        ctx._kin_owner = id(self)
        self._configured = True
        del keep  # the library copied every table

    def run_many(self, n_events: int, first_event: int = 0, seed: int | None = None,
                 return_status: bool = False):
        """Generate ``n_events`` events with global ids ``first_event ...`` on the device.

        Returns ``(vertex [n,3], p4 [n,N,4])`` (+ ``(status, attempts)`` if asked).  Raises
        ``PipelineError`` if any event exhausts ``event_sample_limit`` (reference
        pipeline.py:316-319) unless ``return_status`` is set."""
        if not is_device_samplable(self):
            return self._run_many_host_sampled(n_events, return_status)
        ctx = self.context
        if not self._configured or getattr(ctx, "_kin_owner", None) != id(self):
            self.configure_device()
        seed = self.seed if seed is None else int(seed)
        n_rows = len(self.result)
        p4 = np.empty((n_events, n_rows, 4), dtype=np.float64)
        vertex = np.empty((n_events, 3), dtype=np.float64)
        status = np.empty(n_events, dtype=np.int32)
        attempts = np.empty(n_events, dtype=np.uint32)
        ctx.check(
            ctx.lib.attpc_kin_run(
                ctx.handle, seed, int(first_event), int(n_events), _abi.dptr(p4), _abi.dptr(vertex),
                _abi.iptr(status, _abi.C.c_int32), _abi.iptr(attempts, _abi.C.c_uint32),
            ),
            "attpc_kin_run",
        )
        if return_status:
            return vertex, p4, status, attempts
        if np.any(status != 0):
            raise PipelineError(
                f"Reached Sampling Limit ({self.event_sample_limit} samples) for a single event! "
                "You may have defined an illegal reaction!"
            )
        return vertex, p4

    def _run_many_host_sampled(self, n_events: int, return_status: bool):
        """Pipelines with user-defined Python distributions: parameters are drawn on the host
        with ``sample()``, the 4-vector arithmetic and the allowed-tests run on the device."""
        n_steps = 1 + len(self.decays)
        n_rows = len(self.result)
        p4 = np.empty((n_events, n_rows, 4), dtype=np.float64)
        vertex = np.empty((n_events, 3), dtype=np.float64)
        attempts = np.zeros(n_events, dtype=np.uint32)
        status = np.ones(n_events, dtype=np.int32)
        pending = np.arange(n_events)
        while pending.size:
            attempts[pending] += 1
            over = attempts[pending] > self.event_sample_limit
            if np.any(over):
                if not return_status:
                    raise PipelineError(
                        f"Reached Sampling Limit ({self.event_sample_limit} samples) for a single "
                        "event! You may have defined an illegal reaction!"
                    )
                pending = pending[~over]
                if not pending.size:
                    break
            beam = np.empty(pending.size)
            ex = np.empty((pending.size, n_steps))
            th = np.empty((pending.size, n_steps))
            ph = np.empty((pending.size, n_steps))
            vx = np.empty((pending.size, 3))
            for i in range(pending.size):
                s = self.sample()
                beam[i] = s.beam_energy
                ex[i] = [s.reaction_excitation] + list(s.decay_excitations)
                th[i] = [s.reaction_theta] + list(s.decay_thetas)
                ph[i] = [s.reaction_phi] + list(s.decay_phis)
                vx[i] = s.vertex
            rows, st = device_calculate(self.reaction, self.decays, beam, ex, th, ph, self.context)
            good = st == 0
            p4[pending[good]] = rows[good]
            vertex[pending[good]] = vx[good]
            status[pending[good]] = 0
            pending = pending[~good]
        if return_status:
            return vertex, p4, status, attempts
        return vertex, p4

    # -------------------------------------------------------- reference API ---------
    def sample(self) -> Sample:
        """Draw one parameter set on the host with numpy (reference pipeline.py:232-283);
        kept for API parity and for pipelines with custom Python distributions."""
        projectile_energy = self.beam_energy
        vertex = np.zeros(3)
        tm = self.target_material
        if tm is not None:
            rho = np.abs(self.rng.normal(0.0, tm.rho_sigma))
            theta = self.rng.uniform(0.0, 2.0 * np.pi)
            vertex[0] = rho * np.cos(theta)
            vertex[1] = rho * np.sin(theta)
            vertex[2] = self.rng.uniform(tm.z_range[0], tm.z_range[1])
            loss = tm.material.get_energy_loss(self.reaction.projectile, projectile_energy, vertex[2:])
            projectile_energy = float(np.asarray(projectile_energy - loss).reshape(-1)[0])
        two_pi = np.pi * 2.0
        n = len(self.excitations)
        return Sample(
            beam_energy=projectile_energy,
            reaction_excitation=self.excitations[0].sample(self.rng),
            reaction_theta=self.polar_dists[0].sample(self.rng),
            reaction_phi=self.rng.uniform(0.0, two_pi),
            vertex=vertex,
            decay_excitations=[self.excitations[i].sample(self.rng) for i in range(1, n)],
            decay_thetas=[self.polar_dists[i].sample(self.rng) for i in range(1, n)],
            decay_phis=[self.rng.uniform(0.0, two_pi) for _ in self.decays],
        )

    def run(self) -> tuple[np.ndarray, np.ndarray]:
        """One event: ``(vertex [3] m, result [N,4] px,py,pz,E MeV)`` (reference
        pipeline.py:285-388).  ``result`` is the reused ``self.result`` buffer, as in the
        reference (callers that keep it must copy)."""
        idx = self._next_event - self._buf_first
        if self._buf_p4 is None or idx >= len(self._buf_p4):
            self._buf_first = self._next_event
            self._buf_vertex, self._buf_p4 = self.run_many(self.batch_size, first_event=self._buf_first)
            idx = 0
        self._next_event += 1
        self.result[:] = self._buf_p4[idx]
        return (self._buf_vertex[idx].copy(), self.result)


def run_kinematics_pipeline(pipeline: KinematicsPipeline, n_events: int, output_path: Path,
                            batch_size: int = 262_144) -> None:
    """Generate ``n_events`` and write the reference's kinematics file layout (reference
    pipeline.py:429-495): group ``data`` (attrs n_events, proton_numbers, mass_numbers,
    chunk_size, n_chunks) / ``chunk_i`` (attrs min_event, max_event) / ``event_j`` datasets
    [N,4] with vertex_x/y/z attrs.  Written with h5py when available, otherwise as an
    ``.npz`` with the same content (``attpc_engine_amd.io``)."""
    from ..io import KinematicsFileWriter

    print("------- AT-TPC Simulation Engine (MI355X) -------")
    print(f"Sampling kinematics from reaction: {pipeline}")
    print(f"Running for {n_events} samples.")
    print(f"Output will be written to {output_path}.")
    writer = KinematicsFileWriter(
        Path(output_path), n_events, pipeline.get_proton_numbers(), pipeline.get_mass_numbers(), CHUNK_SIZE
    )
    done = 0
    while done < n_events:
        n = min(batch_size, n_events - done)
        vertex, p4 = pipeline.run_many(n, first_event=done)
        writer.write_batch(done, vertex, p4)
        done += n
    writer.close()
    print("Done.")
    print("----------------------------------------")
