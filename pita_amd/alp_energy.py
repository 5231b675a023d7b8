"""Molecular force-field target (alanine-dipeptide class) on the HIP kernel.

Stands in for ``ALPEnergy`` (pita/src/energies/alp_energy.py:24-149): same call contract --
``energy(samples[B, 3n], return_force=False) -> logp`` (and force), samples in the reference's normalised Cartesian
coordinates (``x * data_normalization_factor`` = nm, 0.1640 for ALDP), log-density ``-E/kT`` at the integrator
temperature.  The reference obtains E from OpenMM (amber14-all + implicit/obc1) through bgflow; here the standard
bonded + nonbonded terms are evaluated by ``pita_ff_logp_force`` from parameter TABLES the caller supplies (e.g.
exported from the OpenMM ``System`` the reference itself serialises, generate_md.py:105-106), including the
GB-OBC1 implicit solvent + ACE surface area of ``implicit/obc1.xml`` when ``gb_radius`` / ``gb_scale`` are given.
No amber parameters ship with the reference tree, so this target is parity-unpinned.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .base_energy_function import BaseMoleculeEnergy

KB_KJ_PER_MOL_K = 8.314462618e-3


def tables_from_openmm_xml(source):
    """Parameter tables from a serialized OpenMM ``System`` (``XmlSerializer.serialize(system)`` -- the file the
    reference itself writes, pita/src/generate_md.py:105-106), parsed with the standard library only: no OpenMM needed
    at run time.  Recognised forces: HarmonicBondForce, HarmonicAngleForce, PeriodicTorsionForce, NonbondedForce
    (particles + exceptions; ``method`` 1 = CutoffNonPeriodic with ``cutoff`` / ``rfDielectric``), GBSAOBCForce, and
    the OBC1 flavour of CustomGBForce that ``implicit/obc1.xml`` creates (per-particle ``charge, or, sr``).  Other
    forces (CMMotionRemover, ...) carry no energy and are skipped; unknown energy-bearing forces raise.
    Returns (tables, options) where options holds cutoff / rf_dielectric / GB dielectrics for ``ForceFieldEnergy``.
    NOT verified against a real OpenMM file in this repository (none ships with the reference): attribute names follow
    OpenMM's serialization proxies."""
    import xml.etree.ElementTree as ET

    root = ET.parse(source).getroot() if not str(source).lstrip().startswith("<") else ET.fromstring(source)
    n = len(root.find("Particles").findall("Particle"))
    t = dict(bond_idx=[], bond_par=[], angle_idx=[], angle_par=[], tors_idx=[], tors_par=[], exc_idx=[], exc_par=[])
    opts = {}
    F = lambda e, k: float(e.attrib[k])
    I = lambda e, k: int(e.attrib[k])
    for force in root.find("Forces").findall("Force"):
        kind = force.attrib.get("type", "")
        if kind == "HarmonicBondForce":
            for b in force.find("Bonds").findall("Bond"):
                t["bond_idx"].append((I(b, "p1"), I(b, "p2")))
                t["bond_par"].append((F(b, "d"), F(b, "k")))
        elif kind == "HarmonicAngleForce":
            for a in force.find("Angles").findall("Angle"):
                t["angle_idx"].append((I(a, "p1"), I(a, "p2"), I(a, "p3")))
                t["angle_par"].append((F(a, "a"), F(a, "k")))
        elif kind == "PeriodicTorsionForce":
            for q in force.find("Torsions").findall("Torsion"):
                t["tors_idx"].append((I(q, "p1"), I(q, "p2"), I(q, "p3"), I(q, "p4")))
                t["tors_par"].append((F(q, "periodicity"), F(q, "phase"), F(q, "k")))
        elif kind == "NonbondedForce":
            parts = force.find("Particles").findall("Particle")
            assert len(parts) == n, "NonbondedForce: particle count mismatch"
            t["charge"] = [F(p_, "q") for p_ in parts]
            t["sigma"] = [F(p_, "sig") for p_ in parts]
            t["epsilon"] = [F(p_, "eps") for p_ in parts]
            exc = force.find("Exceptions")
            for e in (exc.findall("Exception") if exc is not None else []):
                t["exc_idx"].append((I(e, "p1"), I(e, "p2")))
                t["exc_par"].append((F(e, "q"), F(e, "sig"), F(e, "eps")))
            method = int(force.attrib.get("method", "0"))
            if method == 1:
                opts["cutoff"] = float(force.attrib["cutoff"])
                opts["rf_dielectric"] = float(force.attrib.get("rfDielectric", 78.3))
            elif method != 0:
                raise NotImplementedError(f"NonbondedForce method {method}: only NoCutoff / CutoffNonPeriodic are built")
        elif kind == "GBSAOBCForce":
            parts = force.find("Particles").findall("Particle")
            t["gb_radius"] = [F(p_, "r") for p_ in parts]
            t["gb_scale"] = [F(p_, "scale") for p_ in parts]
            opts["gb_solute_dielectric"] = float(force.attrib.get("soluteDielectric", 1.0))
            opts["gb_solvent_dielectric"] = float(force.attrib.get("solventDielectric", 78.5))
            opts["gb_surface_area_factor"] = 4 * np.pi * float(force.attrib.get("surfaceAreaEnergy", 2.25936))
        elif kind == "CustomGBForce":
            names = [e.attrib["name"] for e in force.find("PerParticleParameters").findall("Parameter")]
            if names[:3] != ["charge", "or", "sr"]:
                raise NotImplementedError(f"CustomGBForce with per-particle parameters {names}: only the OBC1 layout is built")
            parts = force.find("Particles").findall("Particle")
            orad = np.array([F(p_, "param2") for p_ in parts])
            srad = np.array([F(p_, "param3") for p_ in parts])
            t["gb_radius"], t["gb_scale"] = (orad + 0.009).tolist(), (srad / orad).tolist()
            glob = {e.attrib["name"]: float(e.attrib["default"]) for e in force.find("GlobalParameters").findall("Parameter")}
            opts["gb_solute_dielectric"] = glob.get("soluteDielectric", 1.0)
            opts["gb_solvent_dielectric"] = glob.get("solventDielectric", 78.5)
        elif kind in ("CMMotionRemover", "MonteCarloBarostat", "AndersenThermostat"):
            continue
        else:
            raise NotImplementedError(f"OpenMM force '{kind}' is not built into pita_ff_logp_force")
    if "charge" not in t:
        raise ValueError("no NonbondedForce in the serialized System")
    return {k: np.asarray(v) for k, v in t.items()}, opts


class ForceFieldEnergy(BaseMoleculeEnergy):
    def __init__(self, tables, n_particles, spatial_dim=3, temperature=300.0, data_normalization_factor=1.0,
                 cutoff=None, rf_dielectric=78.3, device="cuda", is_molecule=True, gb_solute_dielectric=1.0,
                 gb_solvent_dielectric=78.5, gb_surface_area_factor=28.3919551, **kwargs):
        """tables: dict of numpy arrays / tensors: bond_idx[nb,2], bond_par[nb,2], angle_idx[na,3], angle_par[na,2],
        tors_idx[nt,4], tors_par[nt,3], charge[n], sigma[n], epsilon[n], exc_idx[ne,2], exc_par[ne,3]; optional
        gb_radius[n], gb_scale[n] switch the GB-OBC1 implicit solvent on (OpenMM GBSAOBCForce parameters)."""
        assert spatial_dim == 3
        super().__init__(dimensionality=3 * n_particles, n_particles=n_particles, spatial_dim=3, data_path=None,
                         device=device, is_molecule=is_molecule, temperature=temperature, should_normalize=False,
                         data_normalization_factor=data_normalization_factor)
        self.name = "forcefield"
        self.kT = KB_KJ_PER_MOL_K * float(temperature)
        self.length_scale = float(data_normalization_factor)
        self.cutoff, self.rf_dielectric = cutoff, float(rf_dielectric)
        f32 = lambda k, cols: np.ascontiguousarray(np.asarray(tables.get(k, np.zeros((0, cols))), dtype=np.float32).reshape(-1, cols))
        i32 = lambda k, cols: np.ascontiguousarray(np.asarray(tables.get(k, np.zeros((0, cols))), dtype=np.int32).reshape(-1, cols))
        self._t = dict(bond_idx=i32("bond_idx", 2), bond_par=f32("bond_par", 2), angle_idx=i32("angle_idx", 3),
                       angle_par=f32("angle_par", 2), tors_idx=i32("tors_idx", 4), tors_par=f32("tors_par", 3),
                       charge=f32("charge", 1), sigma=f32("sigma", 1), epsilon=f32("epsilon", 1),
                       exc_idx=i32("exc_idx", 2), exc_par=f32("exc_par", 3))
        assert self._t["charge"].shape[0] == n_particles
        self._gb = "gb_radius" in tables
        if self._gb:
            self._t["gb_radius"], self._t["gb_scale"] = f32("gb_radius", 1), f32("gb_scale", 1)
            assert self._t["gb_radius"].shape[0] == n_particles and self._t["gb_scale"].shape[0] == n_particles
        self._gb_par = (float(gb_solute_dielectric), float(gb_solvent_dielectric), float(gb_surface_area_factor))
        self._handle = None

    @classmethod
    def from_openmm_xml(cls, source, temperature=300.0, data_normalization_factor=0.1640, **kw):
        """Target from the serialized OpenMM ``System`` of the reference's ALPEnergy (alp_energy.py:93-100)."""
        tables, opts = tables_from_openmm_xml(source)
        opts.update(kw)
        return cls(tables, n_particles=len(tables["charge"]), temperature=temperature,
                   data_normalization_factor=data_normalization_factor, **opts)

    def _native(self):
        if self._handle is None:
            t = self._t
            P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            cfg = _lib.FfConfig(self.n_particles, len(t["bond_idx"]), P(t["bond_idx"]), P(t["bond_par"]),
                                len(t["angle_idx"]), P(t["angle_idx"]), P(t["angle_par"]),
                                len(t["tors_idx"]), P(t["tors_idx"]), P(t["tors_par"]),
                                P(t["charge"]), P(t["sigma"]), P(t["epsilon"]),
                                len(t["exc_idx"]), P(t["exc_idx"]), P(t["exc_par"]),
                                int(self.cutoff is not None), float(self.cutoff or 0.0), self.rf_dielectric,
                                self.length_scale, self.kT,
                                P(t["gb_radius"]) if self._gb else None, P(t["gb_scale"]) if self._gb else None,
                                *self._gb_par)
            h = ctypes.c_void_p()
            _lib.check(_lib.lib().pita_ff_create(ctypes.byref(h), ctypes.byref(cfg)), "pita_ff_create")
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().pita_ff_destroy(self._handle)
        except Exception:
            pass

    def __call__(self, samples: torch.Tensor, return_force=False):
        x = _lib.dev_tensor(samples, "samples").reshape(-1, self._dimensionality)
        B = x.shape[0]
        logp = torch.empty(B, device=x.device, dtype=torch.float32)
        force = torch.empty_like(x) if return_force else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().pita_ff_logp_force(self._native(), x.data_ptr(), logp.data_ptr(), _lib.ptr(force), B,
                                                     _lib.stream_ptr(x.device)), "pita_ff_logp_force")
        return (logp, force) if return_force else logp


    def fused_descent(self, x, num_steps, dt, noise_scale, sqrt_dt, seed=0, walker_offset=0, step0=0, remove_mean=True,
                      noise=None):
        """``num_steps`` of x <- remove_mean(x + F dt + noise_scale*sqrt_dt*xi) in ONE launch, in place
        (negative_time_descent, sde_integration.py:353-360; pita_ff_descent), bit-identical to the per-step path."""
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().pita_ff_descent(
                self._native(), x.data_ptr(), _lib.ptr(noise), x.shape[0], int(num_steps), float(dt), float(noise_scale),
                float(sqrt_dt), int(seed) & 0xFFFFFFFFFFFFFFFF, int(walker_offset), int(step0), int(bool(remove_mean)),
                _lib.stream_ptr(x.device)), "pita_ff_descent")
        return x


class ALPEnergy(ForceFieldEnergy):
    """Drop-in for ``src.energies.alp_energy.ALPEnergy`` (pita/src/energies/alp_energy.py:41-149): the reference
    constructor's argument names and defaults, the same call contract (``energy(samples, return_force=False)``, samples
    normalised Cartesian coordinates, ``maybe_unnormalize`` = x * data_normalization_factor when ``should_normalize``;
    log-density = -E / kT at the integrator temperature ``temperature`` in kelvin, alp_energy.py:101-105,136), so a Hydra
    ``_target_: pita_amd.alp_energy.ALPEnergy`` can take over ``energy/aldp.yaml``.

    One difference, stated: the reference builds its OpenMM ``System`` from ``pdb_filename`` + amber14-all / implicit/obc1
    (alp_energy.py:86-100); neither OpenMM nor those parameter files exist here, so this class takes the SAME system in
    serialized form -- ``system_xml=`` the file ``XmlSerializer.serialize(system)`` writes (what the reference itself
    does in generate_md.py:105-106).  ``pdb_filename`` / ``atom_encoding_filename`` / ``data_path`` are accepted and
    kept (the sampling path never reads them).  Without ``system_xml`` the constructor raises.

    PARITY UNPINNED: no OpenMM, no amber14 tables and no reference energy values for this system exist in the reference
    tree; the kernel is checked against the repository's own CPU restatement of OpenMM's functional forms only."""

    def __init__(self, data_path=None, pdb_filename=None, atom_encoding_filename="atom_types_ecoding.npy",
                 dimensionality=99, n_particles=33, spatial_dim=3, device="cuda", plot_samples_epoch_period=5,
                 plotting_buffer_sample_size=512, data_normalization_factor=1.0, is_molecule=True, temperature=1.0,
                 should_normalize=True, should_remove_mean=False, device_index=0, debug_train_on_test=False,
                 energy_batch_size=10000, system_xml=None):
        if system_xml is None:
            raise _lib.PitaHipError(
                "ALPEnergy: pass system_xml= (the serialized OpenMM System of the molecule, "
                "XmlSerializer.serialize(system) as in pita/src/generate_md.py:105-106); building it from "
                f"pdb_filename={pdb_filename!r} needs OpenMM + amber14, which are not available to this library")
        tables, opts = tables_from_openmm_xml(system_xml)
        if len(tables["charge"]) != int(n_particles) or int(dimensionality) != int(n_particles) * int(spatial_dim):
            raise ValueError(f"ALPEnergy: the System has {len(tables['charge'])} particles, the configuration says "
                             f"n_particles={n_particles}, dimensionality={dimensionality}")
        super().__init__(tables, n_particles=int(n_particles), spatial_dim=int(spatial_dim), temperature=float(temperature),
                         data_normalization_factor=float(data_normalization_factor) if should_normalize else 1.0,
                         device=device, is_molecule=is_molecule, **opts)
        self.name = "AL"
        self.data_path, self.pdb_path, self.atom_encoding_filename = data_path, pdb_filename, atom_encoding_filename
        self.should_normalize, self.should_remove_mean = bool(should_normalize), bool(should_remove_mean)
        self.data_normalization_factor = float(data_normalization_factor)
        self.debug_train_on_test, self.energy_batch_size = debug_train_on_test, int(energy_batch_size)
        self.device_index = device_index
        self.plot_samples_epoch_period, self.plotting_buffer_sample_size = plot_samples_epoch_period, plotting_buffer_sample_size

    def __call__(self, samples: torch.Tensor, return_force=False):
        """alp_energy.py:122-149: chunks of ``energy_batch_size`` (kept: it bounds the per-launch batch), results detached."""
        x = _lib.dev_tensor(samples, "samples").reshape(-1, self._dimensionality)
        if x.shape[0] <= self.energy_batch_size:
            return super().__call__(x, return_force)
        outs = [super(ALPEnergy, self).__call__(c.contiguous(), return_force) for c in torch.split(x, self.energy_batch_size)]
        if return_force:
            return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        return torch.cat(outs)
