"""Synthetic molecular topologies for the force-field tests and timing tools (no oracle import here)."""
import numpy as np


def synthetic_peptide(n=22, seed=0):
    """A 22-atom branched chain with bonded tables and nonbonded parameters of realistic magnitude (the real amber14
    parameters of alanine dipeptide are outside the reference tree)."""
    rng = np.random.default_rng(seed)
    parent = [-1] + [int(rng.integers(max(0, i - 3), i)) for i in range(1, n)]
    bonds = [(parent[i], i) for i in range(1, n)]
    nbr = {i: set() for i in range(n)}
    for a, b in bonds:
        nbr[a].add(b); nbr[b].add(a)
    angles = sorted({(a, j, c) for j in range(n) for a in nbr[j] for c in nbr[j] if a < c})
    tors = sorted({(a, j, k, d) for (j, k) in bonds + [(b, a) for a, b in bonds] if j < k for a in nbr[j] - {k} for d in nbr[k] - {j} if a != d})
    # equilibrium-ish coordinates: random walk with bond length ~0.15 nm
    pos = np.zeros((n, 3))
    for i in range(1, n):
        v = rng.normal(size=3); v /= np.linalg.norm(v)
        pos[i] = pos[parent[i]] + 0.15 * v
    t = dict(bond_idx=np.array(bonds), bond_par=np.stack([rng.uniform(0.1, 0.16, len(bonds)), rng.uniform(2e5, 4e5, len(bonds))], 1),
             angle_idx=np.array(angles), angle_par=np.stack([rng.uniform(1.7, 2.2, len(angles)), rng.uniform(300, 700, len(angles))], 1),
             tors_idx=np.array(tors), tors_par=np.stack([rng.integers(1, 4, len(tors)).astype(float), rng.choice([0.0, np.pi], len(tors)), rng.uniform(0.5, 8, len(tors))], 1),
             charge=rng.normal(0, 0.35, n), sigma=rng.uniform(0.1, 0.34, n), epsilon=rng.uniform(0.05, 0.7, n))
    exc, par = [], []
    for a, b in bonds:
        exc.append((a, b)); par.append((0, 1, 0))
    for a, j, c in angles:
        exc.append((a, c)); par.append((0, 1, 0))
    seen = set(map(tuple, map(sorted, exc)))
    for a, j, k, d in tors:
        if tuple(sorted((a, d))) not in seen:
            seen.add(tuple(sorted((a, d))))
            exc.append((a, d)); par.append((t["charge"][a] * t["charge"][d] / 1.2, 0.5 * (t["sigma"][a] + t["sigma"][d]), 0.5 * np.sqrt(t["epsilon"][a] * t["epsilon"][d])))
    t["exc_idx"], t["exc_par"] = np.array(exc), np.array(par, dtype=float)
    return t, pos


def openmm_system_xml(t, cutoff=2.0, rf_dielectric=78.3, gb=True):
    """The tables of ``synthetic_peptide`` written the way OpenMM's ``XmlSerializer.serialize(system)`` lays out a
    System (element and attribute names of OpenMM's serialization proxies: HarmonicBondForce, HarmonicAngleForce,
    PeriodicTorsionForce, NonbondedForce with exceptions, the CustomGBForce that implicit/obc1.xml creates,
    CMMotionRemover).  Written by hand from the format's published layout: no OpenMM here to produce a real one."""
    n = len(t["charge"])
    r = lambda v: repr(float(v))
    L = ['<?xml version="1.0" ?>', '<System openmmVersion="8.1" type="System" version="1">',
         '\t<PeriodicBoxVectors>', '\t\t<A x="2" y="0" z="0"/>', '\t\t<B x="0" y="2" z="0"/>', '\t\t<C x="0" y="0" z="2"/>',
         '\t</PeriodicBoxVectors>', '\t<Particles>']
    L += ['\t\t<Particle mass="12.01"/>'] * n
    L += ['\t</Particles>', '\t<Constraints/>', '\t<Forces>']
    L.append('\t\t<Force forceGroup="0" name="HarmonicBondForce" type="HarmonicBondForce" usesPeriodic="0" version="2">')
    L.append('\t\t\t<Bonds>')
    for (a, b), (d, k) in zip(t["bond_idx"], t["bond_par"]):
        L.append(f'\t\t\t\t<Bond d="{r(d)}" k="{r(k)}" p1="{int(a)}" p2="{int(b)}"/>')
    L += ['\t\t\t</Bonds>', '\t\t</Force>']
    L.append('\t\t<Force forceGroup="0" name="HarmonicAngleForce" type="HarmonicAngleForce" usesPeriodic="0" version="2">')
    L.append('\t\t\t<Angles>')
    for (a, b, c), (th, k) in zip(t["angle_idx"], t["angle_par"]):
        L.append(f'\t\t\t\t<Angle a="{r(th)}" k="{r(k)}" p1="{int(a)}" p2="{int(b)}" p3="{int(c)}"/>')
    L += ['\t\t\t</Angles>', '\t\t</Force>']
    L.append('\t\t<Force forceGroup="0" name="PeriodicTorsionForce" type="PeriodicTorsionForce" usesPeriodic="0" version="2">')
    L.append('\t\t\t<Torsions>')
    for (a, b, c, d), (per, ph, k) in zip(t["tors_idx"], t["tors_par"]):
        L.append(f'\t\t\t\t<Torsion k="{r(k)}" p1="{int(a)}" p2="{int(b)}" p3="{int(c)}" p4="{int(d)}" '
                 f'periodicity="{int(per)}" phase="{r(ph)}"/>')
    L += ['\t\t\t</Torsions>', '\t\t</Force>']
    L.append(f'\t\t<Force alpha="0" cutoff="{r(cutoff)}" dispersionCorrection="1" ewaldTolerance=".0005" '
             'exceptionsUsePeriodic="0" forceGroup="0" includeDirectSpace="1" ljAlpha="0" ljnx="0" ljny="0" ljnz="0" '
             f'method="1" name="NonbondedForce" nx="0" ny="0" nz="0" recipForceGroup="-1" rfDielectric="{r(rf_dielectric)}" '
             'switchingDistance="-1" type="NonbondedForce" useSwitchingFunction="0" version="4">')
    L += ['\t\t\t<GlobalParameters/>', '\t\t\t<ParticleOffsets/>', '\t\t\t<ExceptionOffsets/>', '\t\t\t<Particles>']
    for q, sg, ep in zip(t["charge"], t["sigma"], t["epsilon"]):
        L.append(f'\t\t\t\t<Particle eps="{r(ep)}" q="{r(q)}" sig="{r(sg)}"/>')
    L += ['\t\t\t</Particles>', '\t\t\t<Exceptions>']
    for (a, b), (q, sg, ep) in zip(t["exc_idx"], t["exc_par"]):
        L.append(f'\t\t\t\t<Exception eps="{r(ep)}" p1="{int(a)}" p2="{int(b)}" q="{r(q)}" sig="{r(sg)}"/>')
    L += ['\t\t\t</Exceptions>', '\t\t</Force>']
    if gb:
        L.append(f'\t\t<Force cutoff="{r(cutoff)}" forceGroup="0" method="1" name="CustomGBForce" type="CustomGBForce" version="3">')
        L += ['\t\t\t<PerParticleParameters>', '\t\t\t\t<Parameter name="charge"/>', '\t\t\t\t<Parameter name="or"/>',
              '\t\t\t\t<Parameter name="sr"/>', '\t\t\t</PerParticleParameters>', '\t\t\t<GlobalParameters>',
              '\t\t\t\t<Parameter default="78.5" name="solventDielectric"/>',
              '\t\t\t\t<Parameter default="1" name="soluteDielectric"/>', '\t\t\t</GlobalParameters>',
              '\t\t\t<EnergyParameterDerivatives/>', '\t\t\t<Particles>']
        for q, rad, sc in zip(t["charge"], t["gb_radius"], t["gb_scale"]):
            orad = float(rad) - 0.009
            L.append(f'\t\t\t\t<Particle param1="{r(q)}" param2="{r(orad)}" param3="{r(float(sc) * orad)}"/>')
        L += ['\t\t\t</Particles>', '\t\t\t<Exclusions/>', '\t\t\t<Functions/>', '\t\t\t<ComputedValues>',
              '\t\t\t\t<Value expression="select(step(r+sr2-or1), 0.5*(1/L-1/U+0.25*(r-sr2^2/r)*(1/(U^2)-1/(L^2))+0.5*log(L/U)/r), 0);'
              'U=r+sr2;L=max(or1, D);D=abs(r-sr2)" name="I" type="2"/>',
              '\t\t\t\t<Value expression="1/(1/or-tanh(0.8*psi+2.909125*psi^3)/radius);psi=I*or;radius=or+offset; offset=0.009" '
              'name="B" type="0"/>', '\t\t\t</ComputedValues>', '\t\t\t<EnergyTerms>',
              '\t\t\t\t<Term expression="28.3919551*(radius+0.14)^2*(radius/B)^6-0.5*138.935485*(1/soluteDielectric-1/solventDielectric)'
              '*charge^2/B;radius=or+offset; offset=0.009" type="0"/>',
              '\t\t\t\t<Term expression="-138.935485*(1/soluteDielectric-1/solventDielectric)*charge1*charge2/f;'
              'f=sqrt(r^2+B1*B2*exp(-r^2/(4*B1*B2)))" type="2"/>', '\t\t\t</EnergyTerms>', '\t\t</Force>']
    L.append('\t\t<Force forceGroup="0" frequency="1" name="CMMotionRemover" type="CMMotionRemover" version="1"/>')
    L += ['\t</Forces>', '</System>']
    return "\n".join(L) + "\n"


def synthetic_peptide_gb(n=22, seed=0):
    """synthetic_peptide plus GB-OBC1 radii / scale factors of realistic magnitude."""
    t, pos = synthetic_peptide(n, seed)
    rng = np.random.default_rng(seed + 1000)
    t["gb_radius"], t["gb_scale"] = rng.uniform(0.12, 0.19, n), rng.uniform(0.72, 0.85, n)
    return t, pos
