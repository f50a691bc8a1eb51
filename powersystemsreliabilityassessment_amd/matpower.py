"""MATPOWER case files in, study case out — the data format on the input side of the hot path.

The reference starts from ``TestSystem = loadcase('case24_ieee_rts')`` (Montecarlo_nsq_single/nsqMain.m:42, Montecarlo_seq/seqMain.m:31):
a MATPOWER "version 2" case file, i.e. MATLAB source that fills ``mpc.baseMVA``, ``mpc.bus``, ``mpc.gen``, ``mpc.branch``, ``mpc.gencost``.
MATPOWER itself is not part of the reference (un-vendored), so neither its ``loadcase`` nor any case file travels with it; this module
reads that format directly, without MATLAB:

  loadcase(path_or_text)      -> dict(baseMVA, bus, gen, branch, gencost, version)        numeric matrices as numpy arrays
  savecase(mpc, path, name)   the same format back (round trips; how a user exports a case for the reference)
  study_case(mpc, unavail)    -> case24.Case: what nsqMain.m:121-153 + MATPOWER's ext2int / makeBdc prepare for the DC-OPF of every state
                              (loads -> virtual generators with Pmax = 0, Pmin = -Pd, cost 1; real generators cost 0; bus Pd zeroed;
                              b = 1 / (x tap), tap 0 -> 1; consecutive internal bus numbers; the type-3 bus as angle reference)

Column numbers are MATPOWER's (idx_bus / idx_gen / idx_brch, 1-based there, 0-based below).  Only what the DC load-curtailment LP reads is
interpreted; everything else is carried through ``savecase`` untouched.  The study case deliberately keeps EVERY generator and branch row as
a sampled component, like the reference (its state vector has one entry per row of mpc.gen and mpc.branch, mc_sampling.m:24-35); rows that
are out of service in the file (GEN_STATUS <= 0, BR_STATUS == 0) are refused rather than silently dropped, because dropping them would shift
the component numbering of the failure-rate tables.
"""
from __future__ import annotations

import os
import re

import numpy as np

from . import case24

# 0-based column indices (MATPOWER idx_bus, idx_gen, idx_brch)
BUS_I, BUS_TYPE, PD, QD = 0, 1, 2, 3
GEN_BUS, PG, QG, QMAX, QMIN, VG, MBASE, GEN_STATUS, PMAX, PMIN = range(10)
F_BUS, T_BUS, BR_R, BR_X, BR_B, RATE_A, RATE_B, RATE_C, TAP, SHIFT, BR_STATUS = range(11)
REF = 3

_FIELDS = ("bus", "gen", "branch", "gencost")


class MatpowerFormatError(ValueError):
    pass


def _strip_comments(text: str) -> str:
    out = []
    for line in text.splitlines():
        k = line.find("%")
        out.append(line if k < 0 else line[:k])
    return "\n".join(out)


def _matrix(body: str, what: str) -> np.ndarray:
    """The rows of a MATLAB matrix literal: rows end at ';' or at a line break, '...' continues a row."""
    body = body.replace("...\n", " ").replace("...", " ")
    rows = []
    for chunk in re.split(r"[;\n]", body):
        chunk = chunk.replace(",", " ").strip()
        if not chunk:
            continue
        try:
            rows.append([float(tok) for tok in chunk.split()])
        except ValueError as e:
            raise MatpowerFormatError(f"mpc.{what}: cannot read the row {chunk[:60]!r}") from e
    if not rows:
        return np.zeros((0, 0))
    width = {len(r) for r in rows}
    if len(width) != 1:
        raise MatpowerFormatError(f"mpc.{what}: rows of different lengths {sorted(width)}")
    return np.array(rows, dtype=np.float64)


def loadcase(source: str) -> dict:
    """Reads a MATPOWER version-2 case (`function mpc = name` ... `mpc.bus = [ ... ];`).  `source` is a path (with or without `.m`)
    or the text of the file.  Returns dict(name, version, baseMVA, bus, gen, branch, gencost); gencost is None if the file has none."""
    text = source
    if "\n" not in source and ("=" not in source):
        path = source if os.path.exists(source) else source + ".m"
        with open(path) as f:
            text = f.read()
    text = _strip_comments(text)
    m = re.search(r"function\s+mpc\s*=\s*(\w+)", text)
    out = dict(name=m.group(1) if m else "", version="2", gencost=None)
    m = re.search(r"mpc\.version\s*=\s*'([^']*)'", text)
    if m:
        out["version"] = m.group(1)
    if out["version"] != "2":
        raise MatpowerFormatError(f"MATPOWER case format version {out['version']!r} is not supported (version 2 is)")
    m = re.search(r"mpc\.baseMVA\s*=\s*([-+0-9.eE]+)\s*;", text)
    if not m:
        raise MatpowerFormatError("mpc.baseMVA is missing")
    out["baseMVA"] = float(m.group(1))
    for name in _FIELDS:
        m = re.search(r"mpc\." + name + r"\s*=\s*\[(.*?)\]\s*;", text, re.S)
        if m:
            out[name] = _matrix(m.group(1), name)
        elif name != "gencost":
            raise MatpowerFormatError(f"mpc.{name} is missing")
    for name, need in (("bus", 13), ("gen", 10), ("branch", 11)):
        if out[name].shape[1] < need:
            raise MatpowerFormatError(f"mpc.{name} has {out[name].shape[1]} columns, MATPOWER's format has at least {need}")
    return out


def savecase(mpc: dict, path: str, name: str | None = None) -> str:
    """Writes the case as a MATPOWER version-2 file (what `loadcase` reads; %.17g keeps every double)."""
    name = name or mpc.get("name") or os.path.splitext(os.path.basename(path))[0]
    fmt = lambda v: ("%d" % v) if float(v).is_integer() and abs(v) < 1e15 else ("%.17g" % v)
    lines = [f"function mpc = {name}", "mpc.version = '2';", "mpc.baseMVA = %s;" % fmt(mpc["baseMVA"])]
    for field in _FIELDS:
        a = mpc.get(field)
        if a is None:
            continue
        lines.append(f"mpc.{field} = [")
        lines += ["\t" + "\t".join(fmt(v) for v in row) + ";" for row in np.asarray(a, dtype=np.float64)]
        lines.append("];")
    if not path.endswith(".m"):
        path += ".m"
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


def study_case(mpc: dict, unavail, always_up=None, elim_order=None) -> case24.Case:
    """The study case the reference builds from a loaded MATPOWER case before its Monte Carlo loop (nsqMain.m:121-153), in the internal
    form MATPOWER's DC-OPF works on (ext2int: consecutive bus numbers; makeBdc: b = 1 / (x tap)).

    unavail    [ng + nl] failure probabilities, generators then branches, in the row order of mpc.gen / mpc.branch (failprob.m:39)
    always_up  [ng + nl] 1 = never sampled as failed (mc_sampling.m:40-41 forces component 15 of case24_ieee_rts up); default none
    """
    bus, gen, br = (np.asarray(mpc[k], dtype=np.float64) for k in ("bus", "gen", "branch"))
    nb, ng, nl = bus.shape[0], gen.shape[0], br.shape[0]
    ids = bus[:, BUS_I].astype(np.int64)
    if np.unique(ids).size != nb:
        raise MatpowerFormatError("mpc.bus: duplicate bus numbers")
    index = {int(b): i for i, b in enumerate(ids)}                    # ext2int: position in mpc.bus
    ref = np.flatnonzero(bus[:, BUS_TYPE] == REF)
    if ref.size != 1:
        raise MatpowerFormatError(f"mpc.bus: {ref.size} reference (type 3) buses, the DC-OPF needs exactly one")
    if np.any(gen[:, GEN_STATUS] <= 0) or np.any(br[:, BR_STATUS] == 0):
        raise MatpowerFormatError("out-of-service rows in mpc.gen / mpc.branch: remove them (and their failure rates) — every row is a sampled component")
    if np.any(br[:, BR_X] == 0):
        raise MatpowerFormatError("mpc.branch: a branch with x = 0 has no DC model")
    if np.any(br[:, SHIFT] != 0):
        raise MatpowerFormatError("mpc.branch: phase shifters are not modelled (the reference's cases have none)")
    try:
        gen_bus = np.array([index[int(b)] for b in gen[:, GEN_BUS]], dtype=np.int32)
        br_from = np.array([index[int(b)] for b in br[:, F_BUS]], dtype=np.int32)
        br_to = np.array([index[int(b)] for b in br[:, T_BUS]], dtype=np.int32)
    except KeyError as e:
        raise MatpowerFormatError(f"a generator or branch names bus {e.args[0]}, which mpc.bus does not list") from e
    bus_pd = bus[:, PD].copy()
    load_buses, vpmin, vpmax = case24.dispatchable_load_model(bus_pd)         # nsqMain.m:121, 136-148
    tap = np.where(br[:, TAP] == 0, 1.0, br[:, TAP])                          # makeBdc: tap 0 means 1
    unavail = np.ascontiguousarray(unavail, dtype=np.float64).ravel()
    if unavail.size != ng + nl:
        raise ValueError(f"unavail has {unavail.size} entries, the case has {ng} generator rows + {nl} branches")
    up = np.zeros(ng + nl, dtype=np.uint8) if always_up is None else np.ascontiguousarray(always_up, dtype=np.uint8).ravel()
    if up.size != ng + nl:
        raise ValueError("always_up must have one entry per generator row and branch")
    return case24.Case(
        base_mva=float(mpc["baseMVA"]), nb=int(nb), ng=int(ng), nl=int(nl), nd=int(load_buses.size), ref_bus=int(ref[0]),
        bus_pd=bus_pd, inj_bus=np.concatenate([gen_bus, load_buses]).astype(np.int32),
        inj_pmin=np.concatenate([gen[:, PMIN], vpmin]), inj_pmax=np.concatenate([gen[:, PMAX], vpmax]),
        inj_cost=np.concatenate([np.zeros(ng), np.ones(load_buses.size)]),        # nsqMain.m:128, 132-133
        br_from=br_from, br_to=br_to, br_b=1.0 / (br[:, BR_X] * tap), br_rate=br[:, RATE_A].copy(),
        unavail=unavail, always_up=up, total_load=float(bus_pd.sum()),              # nsqMain.m:125
        elim_order=None if elim_order is None else np.ascontiguousarray(elim_order, dtype=np.int32))


def mpc_from_arrays(base_mva, bus_pd, gen_bus, gen_pmax, gen_pmin, br_from, br_to, br_x, br_rate, br_tap, ref_bus, bus_numbers=None, name="") -> dict:
    """A MATPOWER case from the plain arrays of this package's case modules (1-based bus numbers as in the case files): what
    `savecase` needs to hand a case of this package to the reference."""
    nb = len(bus_pd)
    num = np.arange(1, nb + 1) if bus_numbers is None else np.asarray(bus_numbers)
    bus = np.zeros((nb, 13)); bus[:, BUS_I] = num; bus[:, BUS_TYPE] = 1; bus[:, PD] = bus_pd
    bus[:, 6] = 1; bus[:, 7] = 1.0; bus[:, 9] = 138.0; bus[:, 10] = 1; bus[:, 11] = 1.05; bus[:, 12] = 0.95
    bus[list(num).index(ref_bus), BUS_TYPE] = REF
    ng = len(gen_bus)
    gen = np.zeros((ng, 21)); gen[:, GEN_BUS] = gen_bus; gen[:, VG] = 1.0; gen[:, MBASE] = base_mva; gen[:, GEN_STATUS] = 1
    gen[:, PMAX] = gen_pmax; gen[:, PMIN] = gen_pmin
    bus[np.isin(num, gen_bus) & (bus[:, BUS_TYPE] == 1), BUS_TYPE] = 2
    nl = len(br_from)
    br = np.zeros((nl, 13)); br[:, F_BUS] = br_from; br[:, T_BUS] = br_to; br[:, BR_X] = br_x
    br[:, RATE_A] = br_rate; br[:, RATE_B] = br_rate; br[:, RATE_C] = br_rate; br[:, TAP] = br_tap; br[:, BR_STATUS] = 1; br[:, 11] = -360; br[:, 12] = 360
    gencost = np.tile([2.0, 0, 0, 3, 0, 0, 0], (ng, 1))
    return dict(name=name, version="2", baseMVA=float(base_mva), bus=bus, gen=gen, branch=br, gencost=gencost)
