"""Dependency-free SMILES -> (x, edge_index, edge_attr) featuriser for the hot path's integer inputs.

Stands in, for the organic subset, for ``smiles2graph`` of ``/root/reference/gnnepcsaft/data/ogb_utils.py:37-147`` (ogb
1.3.6's featuriser on top of RDKit), which cannot run here: RDKit is not installable offline.  Same output contract:
``node_feat int64[n, 9]`` (atomic number, chirality tag, total degree, formal charge, total Hs, radical electrons,
hybridisation, is-aromatic, is-in-ring: indices into the lists at ogb_utils.py:8-23), ``edge_feat int64[e, 3]`` (bond
type, bond stereo, is-conjugated: ogb_utils.py:24-33), ``edge_index int64[2, e]`` with both directions of every bond
adjacent (ogb_utils.py:125-129), ``edge_index[2, 0]`` for a single atom (ogb_utils.py:137-139); hydrogens implicit.

PARITY UNPINNED: the perception rules below restate RDKit's published algorithms from memory (SMILES valence model,
its aromaticity model on SSSR-style rings and fused pairs, ``setConjugation`` / ``setHybridization`` of ConjugHybrid.cpp,
'@' = counter-clockwise with the ring-closure permutation rule, E/Z from directional bonds with CIP-style ranks); they
are pinned only by known answers for common molecules (tests/test_featurize_cpu.py).  Out of scope: atoms outside
B C N O F Si P S Cl Br I Se As + bracket atoms of any element with default handling, radicals, allenes / square-planar
stereo, tautomer / charge normalisation, and RDKit's sanitisation errors (an invalid valence is accepted as written).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

# ---- vocabularies (ogb_utils.py:8-33): index = feature value -----------------------------------------------------
_CHIRAL = {"CHI_UNSPECIFIED": 0, "CHI_TETRAHEDRAL_CW": 1, "CHI_TETRAHEDRAL_CCW": 2}
_HYB = {"SP": 0, "SP2": 1, "SP3": 2, "SP3D": 3, "SP3D2": 4, "misc": 5}
_BOND = {1.0: 0, 2.0: 1, 3.0: 2, 1.5: 3}
_STEREO = {"STEREONONE": 0, "STEREOZ": 1, "STEREOE": 2}

_SYMBOLS = ["H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne", "Na", "Mg", "Al", "Si", "P", "S", "Cl", "Ar", "K", "Ca",
            "Sc", "Ti", "V", "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr", "Rb", "Sr", "Y",
            "Zr", "Nb", "Mo", "Tc", "Ru", "Rh", "Pd", "Ag", "Cd", "In", "Sn", "Sb", "Te", "I", "Xe", "Cs", "Ba", "La", "Ce",
            "Pr", "Nd", "Pm", "Sm", "Eu", "Gd", "Tb", "Dy", "Ho", "Er", "Tm", "Yb", "Lu", "Hf", "Ta", "W", "Re", "Os", "Ir",
            "Pt", "Au", "Hg", "Tl", "Pb", "Bi", "Po", "At", "Rn", "Fr", "Ra", "Ac", "Th", "Pa", "U", "Np", "Pu", "Am", "Cm",
            "Bk", "Cf", "Es", "Fm", "Md", "No", "Lr", "Rf", "Db", "Sg", "Bh", "Hs", "Mt", "Ds", "Rg", "Cn", "Nh", "Fl", "Mc",
            "Lv", "Ts", "Og"]
_Z = {s: i + 1 for i, s in enumerate(_SYMBOLS)}
_ORGANIC = ("Cl", "Br", "B", "C", "N", "O", "P", "S", "F", "I")
_AROMATIC_ORGANIC = {"b": "B", "c": "C", "n": "N", "o": "O", "p": "P", "s": "S"}
_AROMATIC_BRACKET = {"se": "Se", "as": "As", **_AROMATIC_ORGANIC}
# SMILES / RDKit default valences (lowest that fits is used for implicit hydrogens)
_VALENCES = {5: (3,), 6: (4,), 7: (3, 5), 8: (2,), 9: (1,), 14: (4,), 15: (3, 5), 16: (2, 4, 6), 17: (1,), 33: (3, 5),
             34: (2, 4, 6), 35: (1,), 53: (1,)}
_OUTER = {1: 1, 5: 3, 6: 4, 7: 5, 8: 6, 9: 7, 14: 4, 15: 5, 16: 6, 17: 7, 33: 5, 34: 6, 35: 7, 53: 7}


class _Atom:
    __slots__ = ("z", "aromatic", "charge", "hcount", "bracket", "chiral", "nbrs", "smiles_slots", "idx")

    def __init__(self, z, aromatic=False, charge=0, hcount=None, bracket=False, chiral=0):
        self.z, self.aromatic, self.charge, self.hcount, self.bracket, self.chiral = z, aromatic, charge, hcount, bracket, chiral
        self.nbrs: List[int] = []          # bond ids in creation order (RDKit's adjacency order)
        self.smiles_slots: List[object] = []  # neighbours in order of appearance in the string: bond id | ("rc", digit) | "H"
        self.idx = -1


class _Bond:
    __slots__ = ("a", "b", "order", "dir", "aromatic", "conj", "stereo", "in_ring")

    def __init__(self, a, b, order, direction=None):
        self.a, self.b, self.order, self.dir = a, b, order, direction
        self.aromatic = order == 1.5
        self.conj, self.stereo, self.in_ring = False, "STEREONONE", False

    def other(self, i):
        return self.b if i == self.a else self.a


# ------------------------------------------------------------------------------------------------------------------
# parsing
# ------------------------------------------------------------------------------------------------------------------
def _parse(smiles: str) -> Tuple[List[_Atom], List[_Bond]]:
    atoms: List[_Atom] = []
    bonds: List[_Bond] = []
    stack: List[int] = []
    prev: Optional[int] = None
    pending: Optional[Tuple[float, Optional[str]]] = None   # explicit bond symbol waiting for its second atom
    ring: Dict[int, Tuple[int, Optional[Tuple[float, Optional[str]]], int]] = {}
    i, n = 0, len(smiles)

    def add_atom(atom: _Atom):
        nonlocal prev, pending
        atom.idx = len(atoms)
        atoms.append(atom)
        if prev is not None:
            order, direction = pending if pending is not None else (None, None)
            if order is None:
                order = 1.5 if (atoms[prev].aromatic and atom.aromatic) else 1.0
            if order == 0.0:  # '.' : no bond
                pass
            else:
                bid = len(bonds)
                bonds.append(_Bond(prev, atom.idx, order, direction))
                atoms[prev].nbrs.append(bid)
                atom.nbrs.append(bid)
                atoms[prev].smiles_slots.append(bid)
                atom.smiles_slots.insert(0, bid)
        pending = None
        prev = atom.idx

    while i < n:
        ch = smiles[i]
        if ch == "[":
            j = smiles.index("]", i)
            atoms_before = len(atoms)
            add_atom(_bracket_atom(smiles[i + 1:j]))
            a = atoms[atoms_before]
            if a.hcount and a.chiral:
                # the implicit hydrogen of [C@H] sits right after the preceding atom in the neighbour order
                pos = 1 if (a.smiles_slots and not isinstance(a.smiles_slots[0], str)) else 0
                a.smiles_slots.insert(pos, "H")
            i = j + 1
        elif ch in "-=#:/\\":
            pending = {"-": (1.0, None), "=": (2.0, None), "#": (3.0, None), ":": (1.5, None), "/": (1.0, "/"),
                       "\\": (1.0, "\\")}[ch]
            i += 1
        elif ch == ".":
            pending = (0.0, None)
            i += 1
        elif ch == "(":
            stack.append(prev)
            i += 1
        elif ch == ")":
            prev = stack.pop()
            i += 1
        elif ch.isdigit() or ch == "%":
            if ch == "%":
                num, i = int(smiles[i + 1:i + 3]), i + 3
            else:
                num, i = int(ch), i + 1
            if num in ring:
                a0, p0, slot = ring.pop(num)
                spec = pending if pending is not None else p0
                order, direction = spec if spec is not None else (None, None)
                if order is None:
                    order = 1.5 if (atoms[a0].aromatic and atoms[prev].aromatic) else 1.0
                if pending is None and p0 is not None and p0[1] is not None:
                    direction = p0[1]
                bid = len(bonds)
                # the bond is created now: last in both atoms' adjacency, whatever its position in the string
                bonds.append(_Bond(a0, prev, order, direction))
                atoms[a0].nbrs.append(bid)
                atoms[prev].nbrs.append(bid)
                atoms[a0].smiles_slots[slot] = bid
                atoms[prev].smiles_slots.append(bid)
                pending = None
            else:
                atoms[prev].smiles_slots.append(("rc", num))
                ring[num] = (prev, pending, len(atoms[prev].smiles_slots) - 1)
                pending = None
        else:
            two = smiles[i:i + 2]
            if two in ("Cl", "Br"):
                add_atom(_Atom(_Z[two]))
                i += 2
            elif ch in _AROMATIC_ORGANIC:
                add_atom(_Atom(_Z[_AROMATIC_ORGANIC[ch]], aromatic=True))
                i += 1
            elif ch in _ORGANIC:
                add_atom(_Atom(_Z[ch]))
                i += 1
            else:
                raise ValueError(f"SMILES is not valid: unexpected {ch!r} at {i} in {smiles!r}")
    if ring or stack:
        raise ValueError("SMILES is not valid: unclosed ring or branch")
    return atoms, bonds


def _bracket_atom(body: str) -> _Atom:
    k = 0
    while k < len(body) and body[k].isdigit():
        k += 1  # isotope: not a feature
    sym = None
    for ln in (2, 1):
        cand = body[k:k + ln]
        if cand in _AROMATIC_BRACKET and (ln == 2 or cand.islower()):
            sym, arom = _AROMATIC_BRACKET[cand], True
            break
        if cand in _Z:
            sym, arom = cand, False
            break
    if sym is None:
        raise ValueError(f"SMILES is not valid: bracket atom [{body}]")
    k += ln
    chiral = 0
    if body[k:k + 2] == "@@":
        chiral, k = 1, k + 2   # clockwise
    elif body[k:k + 1] == "@":
        chiral, k = 2, k + 1   # counter-clockwise
    h = 0
    if body[k:k + 1] == "H":
        k += 1
        d = ""
        while k < len(body) and body[k].isdigit():
            d, k = d + body[k], k + 1
        h = int(d) if d else 1
    charge = 0
    while k < len(body) and body[k] in "+-":
        sign = 1 if body[k] == "+" else -1
        k += 1
        d = ""
        while k < len(body) and body[k].isdigit():
            d, k = d + body[k], k + 1
        charge += sign * (int(d) if d else 1)
    return _Atom(_Z[sym], aromatic=arom, charge=charge, hcount=h, bracket=True, chiral=chiral)


# ------------------------------------------------------------------------------------------------------------------
# perception
# ------------------------------------------------------------------------------------------------------------------
def _explicit_valence(atom: _Atom, bonds: List[_Bond]) -> float:
    return sum(bonds[b].order for b in atom.nbrs)


def _implicit_hs(atoms: List[_Atom], bonds: List[_Bond]) -> None:
    for a in atoms:
        if a.bracket:
            continue
        vals = _VALENCES.get(a.z)
        if vals is None:
            a.hcount = 0
            continue
        ev = _explicit_valence(a, bonds)
        if a.aromatic:  # SMILES rule: the aromatic system takes one valence beyond the sigma bonds
            ev = len(a.nbrs) + 1 if any(bonds[b].aromatic for b in a.nbrs) else ev
            ev += sum(bonds[b].order - 1 for b in a.nbrs if not bonds[b].aromatic)
        ev = int(round(ev))
        if a.aromatic:
            # RDKit (Atom::calcImplicitValence): an aromatic atom is only ever completed to its LOWEST default valence --
            # thiophene's s (explicit 3 > 2), an N-substituted pyrrole n (4 > 3) and furan's o get no hydrogen.  (Found
            # by tests/test_featurize_inchi_cpu.py: every thiophene of the reference's Esper table had one H too many
            # against its InChI formula while the rule tried S's next valence, 4.)
            a.hcount = max(vals[0] - ev, 0)
        else:
            a.hcount = next((v - ev for v in vals if v >= ev), 0)


def _rings(atoms: List[_Atom], bonds: List[_Bond]) -> List[List[int]]:
    """Ring membership (bridge detection) + a smallest-rings set: for every ring bond the shortest cycle through it."""
    n = len(atoms)
    adj = [[(bonds[b].other(i), b) for b in atoms[i].nbrs] for i in range(n)]
    disc, low, timer = [-1] * n, [0] * n, [0]
    bridge = [False] * len(bonds)
    import sys
    sys.setrecursionlimit(max(10000, 4 * n + 100))

    def dfs(u, pb):
        disc[u] = low[u] = timer[0]
        timer[0] += 1
        for v, b in adj[u]:
            if b == pb:
                continue
            if disc[v] < 0:
                dfs(v, b)
                low[u] = min(low[u], low[v])
                if low[v] > disc[u]:
                    bridge[b] = True
            else:
                low[u] = min(low[u], disc[v])

    for s in range(n):
        if disc[s] < 0:
            dfs(s, -1)
    rings, seen = [], set()
    for bid, bd in enumerate(bonds):
        bd.in_ring = not bridge[bid]
        if bridge[bid]:
            continue
        # shortest path a -> b avoiding this bond (BFS over ring bonds)
        from collections import deque
        prev = {bd.a: None}
        dq = deque([bd.a])
        while dq and bd.b not in prev:
            u = dq.popleft()
            for v, b in adj[u]:
                if b == bid or bridge[b] or v in prev:
                    continue
                prev[v] = u
                dq.append(v)
        path, u = [], bd.b
        while u is not None:
            path.append(u)
            u = prev[u]
        key = frozenset(path)
        if key not in seen:
            seen.add(key)
            rings.append(path)
    return rings


def _pi_electrons(i: int, atoms: List[_Atom], bonds: List[_Bond], system: set) -> Optional[int]:
    """Electrons atom i offers to a ring system (RDKit's default aromaticity model, common cases); None = cannot be
    aromatic."""
    a = atoms[i]
    if a.z not in (5, 6, 7, 8, 15, 16, 33, 34):
        return None
    dbl_in = [b for b in a.nbrs if bonds[b].order == 2.0 and bonds[b].other(i) in system]
    dbl_out = [b for b in a.nbrs if bonds[b].order == 2.0 and bonds[b].other(i) not in system]
    if any(bonds[b].order == 3.0 for b in a.nbrs):
        return None
    if dbl_in:
        return 1 if len(dbl_in) == 1 and not dbl_out else None
    if dbl_out:
        # exocyclic double bond: only towards a more electronegative atom (C=O, C=N, C=S): the ring atom gives 0
        o = atoms[bonds[dbl_out[0]].other(i)]
        return 0 if (a.z == 6 and o.z in (7, 8, 16)) else None
    sigma = len(a.nbrs) + (a.hcount or 0)
    if a.z in (7, 15, 33):
        return 2 if (sigma == 3 and a.charge == 0) else (None if a.charge == 0 else (1 if a.charge > 0 else 2))
    if a.z in (8, 16, 34):
        return 2 if (sigma == 2 and a.charge == 0) else None
    if a.z == 6:
        return 2 if a.charge < 0 else (0 if a.charge > 0 else None)
    if a.z == 5:
        return 0 if sigma == 3 else None
    return None


def _perceive_aromaticity(atoms: List[_Atom], bonds: List[_Bond], rings: List[List[int]]) -> None:
    """Kekule input: rings (and pairs / triples of fused rings) whose atoms all donate and hold 4n+2 electrons."""
    if any(a.aromatic for a in atoms):
        for b in bonds:  # aromatic input is trusted: bonds between aromatic ring atoms written without a symbol
            if b.order == 1.5:
                b.aromatic = True
        return
    cand = [r for r in rings if 5 <= len(r) <= 7 or len(r) > 7]
    ring_sets = [set(r) for r in cand]
    systems = [({k}, s) for k, s in enumerate(ring_sets)]
    for k in range(len(cand)):           # fused pairs and triples (naphthalene, azulene, indole, anthracene ...)
        for m in range(k + 1, len(cand)):
            if len(ring_sets[k] & ring_sets[m]) >= 2:
                systems.append(({k, m}, ring_sets[k] | ring_sets[m]))
                for q in range(m + 1, len(cand)):
                    if len((ring_sets[k] | ring_sets[m]) & ring_sets[q]) >= 2:
                        systems.append(({k, m, q}, ring_sets[k] | ring_sets[m] | ring_sets[q]))
    aromatic_rings = set()
    for members, atoms_in in systems:
        if members <= aromatic_rings:
            continue
        total = 0
        for i in atoms_in:
            e = _pi_electrons(i, atoms, bonds, atoms_in)
            if e is None:
                total = None
                break
            total += e
        if total is not None and total >= 2 and (total - 2) % 4 == 0:
            aromatic_rings |= members
    for k in aromatic_rings:
        r = cand[k]
        rs = ring_sets[k]
        for i in r:
            atoms[i].aromatic = True
        for b in bonds:
            if b.a in rs and b.b in rs and b.in_ring and _adjacent_in_ring(r, b.a, b.b):
                b.aromatic, b.order = True, 1.5


def _adjacent_in_ring(ring: List[int], a: int, b: int) -> bool:
    n = len(ring)
    return any((ring[k] == a and ring[(k + 1) % n] == b) or (ring[k] == b and ring[(k + 1) % n] == a) for k in range(n))


def _count_atom_elec(a: _Atom, bonds: List[_Bond]) -> int:
    """RDKit ConjugHybrid.cpp countAtomElec: electrons an atom can put into a pi system (-1: too many substituents)."""
    vals = _VALENCES.get(a.z)
    dv = vals[0] if vals else 0
    if dv <= 1:
        return 0
    degree = len(a.nbrs) + (a.hcount or 0)
    if degree > 3:
        return -1
    nlp = max(_OUTER.get(a.z, 0) - dv - a.charge, 0)
    res = (dv - degree) + nlp
    if res > 1:
        # an incident bond of order > 2 (or two double bonds): only one electron goes into a given pi system
        unsat = int(round(sum(_kekule_order(bonds[b]) for b in a.nbrs))) - len(a.nbrs)
        if unsat > 1:
            res = 1
    return res


def _kekule_order(b: _Bond) -> float:
    return 1.5 if b.aromatic else b.order


def _conjugation(atoms: List[_Atom], bonds: List[_Bond]) -> None:
    """RDKit MolOps::setConjugation: aromatic bonds are conjugated; a multiple bond and a neighbouring bond whose far
    atom can donate into the pi system (<= 3 substituents, electrons available) are conjugated."""
    for b in bonds:
        b.conj = b.aromatic
    for i, at in enumerate(atoms):
        sbo = len(at.nbrs) + (at.hcount or 0)
        if sbo < 2 or sbo > 3:
            continue
        for b1 in at.nbrs:
            if _kekule_order(bonds[b1]) < 1.5:
                continue
            for b2 in at.nbrs:
                if b1 == b2:
                    continue
                at2 = atoms[bonds[b2].other(i)]
                if len(at2.nbrs) + (at2.hcount or 0) > 3:
                    continue
                if _count_atom_elec(at2, bonds) > 0:
                    bonds[b1].conj = True
                    bonds[b2].conj = True


def _hybridization(a: _Atom, bonds: List[_Bond]) -> int:
    """RDKit setHybridization: steric number = total degree + lone pairs; 4 drops to SP2 on a conjugated atom with at
    most 3 substituents (the hydroxyl O of a carboxylic acid, an amide N, an aniline N)."""
    deg = len(a.nbrs) + (a.hcount or 0)
    nouter = _OUTER.get(a.z)
    if nouter is None:
        return _HYB["misc"] if deg > 6 else (_HYB["SP3"] if deg == 4 else _HYB["misc"])
    total_valence = int(round(sum(_kekule_order(bonds[b]) for b in a.nbrs) + 1e-9)) + (a.hcount or 0)
    if a.aromatic and any(bonds[b].aromatic for b in a.nbrs):
        arom = sum(1 for b in a.nbrs if bonds[b].aromatic)
        total_valence = int(sum(bonds[b].order for b in a.nbrs if not bonds[b].aromatic)) + arom + 1 + (a.hcount or 0)
        if a.z in (7, 15) and deg == 3 or a.z in (8, 16, 34):
            total_valence -= 1  # lone-pair donors (pyrrole N, furan O): the extra aromatic valence is their lone pair
    free = nouter - (total_valence + a.charge)
    norbs = deg + max(free, 0) // 2
    if norbs <= 1:
        return _HYB["misc"]  # RDKit: S
    if norbs == 2:
        return _HYB["SP"]
    if norbs == 3:
        return _HYB["SP2"]
    if norbs == 4:
        return _HYB["SP3"] if (deg > 3 or not any(bonds[b].conj for b in a.nbrs)) else _HYB["SP2"]
    return _HYB["SP3D"] if norbs == 5 else (_HYB["SP3D2"] if norbs == 6 else _HYB["misc"])


def _ranks(atoms: List[_Atom], bonds: List[_Bond]) -> List[int]:
    """Symmetry classes by iterative refinement of (Z, degree, Hs, charge, sorted neighbour classes): decides whether a
    stereo centre has four different substituents and stands in for CIP priorities in the E/Z assignment."""
    inv = [(a.z, len(a.nbrs), a.hcount or 0, a.charge, a.aromatic) for a in atoms]
    cls = _dense(inv)
    for _ in range(len(atoms)):
        nxt = _dense([(cls[i], tuple(sorted((cls[bonds[b].other(i)], _kekule_order(bonds[b])) for b in a.nbrs)))
                      for i, a in enumerate(atoms)])
        if nxt == cls:
            break
        cls = nxt
    return cls


def _dense(keys) -> List[int]:
    order = {k: r for r, k in enumerate(sorted(set(keys)))}
    return [order[k] for k in keys]


def _cip_key(start: int, frm: int, atoms: List[_Atom], bonds: List[_Bond], depth: int = 4):
    """Breadth-first atomic-number signature of the substituent ``start`` seen from ``frm`` (duplicate atoms for
    multiple bonds), compared lexicographically: a practical subset of the CIP sequence rules."""
    key, frontier = [], [(start, frm)]
    for _ in range(depth):
        layer, nxt = [], []
        for u, p in frontier:
            layer.append(atoms[u].z)
            for b in atoms[u].nbrs:
                v = bonds[b].other(u)
                if v == p:
                    continue
                nxt.append((v, u))
                for _dup in range(int(_kekule_order(bonds[b]) + 0.5) - 1):
                    nxt.append((v, u))
            for _h in range(atoms[u].hcount or 0):
                layer.append(0)
        key.append(tuple(sorted(layer, reverse=True)))
        frontier = nxt
        if not frontier:
            break
    return tuple(key)


def _chirality(atoms: List[_Atom], bonds: List[_Bond], cls: List[int]) -> List[int]:
    out = [0] * len(atoms)
    for i, a in enumerate(atoms):
        if not a.chiral:
            continue
        subs = [cls[bonds[b].other(i)] for b in a.nbrs] + [-1] * (a.hcount or 0)
        if len(subs) != 4 or len(set(subs)) != 4:
            continue  # not a tetrahedral stereo centre: RDKit drops the tag
        # '@' / '@@' refer to the neighbours in the order they appear in the string; RDKit's tag refers to its bond
        # order (creation order: ring-closure bonds come last).  An odd permutation between the two flips the tag.
        # An implicit hydrogen takes its place in the string's order ([C@H]: right after the preceding atom, or first
        # if the atom opens the string) and the LAST place in RDKit's.
        ref = list(a.nbrs) + ["H"] * (a.hcount or 0)
        slots = [s for s in a.smiles_slots if isinstance(s, (int, str))]
        if len(slots) != len(ref):
            continue
        perm, used = [], set()
        for s_ in slots:
            k = next(q for q, r in enumerate(ref) if r == s_ and q not in used)
            used.add(k)
            perm.append(k)
        inv = sum(1 for x in range(len(perm)) for y in range(x + 1, len(perm)) if perm[x] > perm[y])
        tag = a.chiral  # 1 = '@@' clockwise, 2 = '@' counter-clockwise
        if inv % 2 == 1:
            tag = 3 - tag
        out[i] = tag
    return out


def _bond_stereo(atoms: List[_Atom], bonds: List[_Bond], cls: List[int]) -> None:
    for b in bonds:
        if b.order != 2.0 or b.aromatic:
            continue
        ends = []
        for me, other in ((b.a, b.b), (b.b, b.a)):
            subs = [bb for bb in atoms[me].nbrs if bonds[bb] is not b]
            marked = [bb for bb in subs if bonds[bb].dir is not None]
            if not marked or len(subs) + (atoms[me].hcount or 0) < 2:
                ends = None
                break
            if len(subs) == 2 and cls[bonds[subs[0]].other(me)] == cls[bonds[subs[1]].other(me)]:
                ends = None  # two identical substituents: no stereo
                break
            m = marked[0]
            nb = bonds[m].other(me)
            # "up" relative to the double-bond atom: '/' written before the atom (neighbour first) points up from the
            # neighbour to the atom, i.e. the neighbour is BELOW; written after the atom the neighbour is ABOVE
            written_nb_first = bonds[m].a == nb
            up = (bonds[m].dir == "/") != written_nb_first
            # is the marked neighbour the higher-priority substituent on this end?
            keys = sorted(((_cip_key(bonds[s].other(me), me, atoms, bonds), s) for s in subs), reverse=True)
            top = keys[0][1] if (len(keys) == 1 or keys[0][0] != keys[1][0]) else None
            if top is None:
                ends = None
                break
            ends.append(up if top == m else (not up))
        if ends is None:
            continue
        b.stereo = "STEREOZ" if ends[0] == ends[1] else "STEREOE"


# ------------------------------------------------------------------------------------------------------------------
# public API (ogb_utils.py:37-147)
# ------------------------------------------------------------------------------------------------------------------
def smiles2graph(smiles_string: str) -> dict:
    """SMILES -> ``{"edge_index", "edge_feat", "node_feat", "num_nodes"}`` with the reference's dtypes and orderings
    (ogb_utils.py:92-147).  Raises ``ValueError("SMILES is not valid")`` on a string that does not parse."""
    try:
        atoms, bonds = _parse(smiles_string)
    except (ValueError, IndexError, KeyError) as e:
        raise ValueError("SMILES is not valid") from e
    if not atoms:
        raise ValueError("SMILES is not valid")
    # explicit [H] atoms bonded to a heavy atom are folded into its hydrogen count (RDKit's default RemoveHs)
    keep = [not (a.z == 1 and len(a.nbrs) == 1 and a.charge == 0 and atoms[bonds[a.nbrs[0]].other(i)].z != 1)
            for i, a in enumerate(atoms)]
    if not all(keep) and any(keep):
        for i, a in enumerate(atoms):
            if not keep[i]:
                heavy = atoms[bonds[a.nbrs[0]].other(i)]
                if heavy.bracket:
                    heavy.hcount = (heavy.hcount or 0) + 1
                heavy.nbrs = [b for b in heavy.nbrs if b != a.nbrs[0]]
                heavy.smiles_slots = ["H" if s == a.nbrs[0] else s for s in heavy.smiles_slots]
        remap, new_atoms = {}, []
        for i, a in enumerate(atoms):
            if keep[i]:
                remap[i] = len(new_atoms)
                new_atoms.append(a)
        old_bonds, bonds, bmap = bonds, [], {}
        for bid, b in enumerate(old_bonds):
            if keep[b.a] and keep[b.b]:
                bmap[bid] = len(bonds)
                b.a, b.b = remap[b.a], remap[b.b]
                bonds.append(b)
        for a in new_atoms:
            a.nbrs = [bmap[b] for b in a.nbrs if b in bmap]
            a.smiles_slots = [bmap.get(s, "H") if isinstance(s, int) else s for s in a.smiles_slots]
        atoms = new_atoms
    rings = _rings(atoms, bonds)
    _implicit_hs(atoms, bonds)
    _perceive_aromaticity(atoms, bonds, rings)
    _conjugation(atoms, bonds)
    cls = _ranks(atoms, bonds)
    chir = _chirality(atoms, bonds, cls)
    _bond_stereo(atoms, bonds, cls)
    in_ring = [any(bonds[b].in_ring for b in a.nbrs) for a in atoms]

    def clip(v, n):  # safe_index: anything outside the list goes to the last ("misc") slot
        return v if 0 <= v < n - 1 else n - 1

    x = np.zeros((len(atoms), 9), dtype=np.int64)
    for i, a in enumerate(atoms):
        h = a.hcount or 0
        x[i] = [clip(a.z - 1, 119), chir[i], clip(len(a.nbrs) + h, 12), clip(a.charge + 5, 12), clip(h, 10), 0,
                _hybridization(a, bonds), int(a.aromatic), int(in_ring[i])]
    if bonds:
        ei, ef = [], []
        for b in bonds:
            feat = [_BOND.get(_kekule_order(b), 4), _STEREO[b.stereo], int(b.conj)]
            ei += [(b.a, b.b), (b.b, b.a)]
            ef += [feat, feat]
        edge_index = np.array(ei, dtype=np.int64).T
        edge_attr = np.array(ef, dtype=np.int64)
    else:
        edge_index = np.empty((2, 0), dtype=np.int64)
        edge_attr = np.empty((0, 3), dtype=np.int64)
    return {"edge_index": edge_index, "edge_feat": edge_attr, "node_feat": x, "num_nodes": len(x)}


def from_smiles(smiles: str, **labels):
    """``Data(x, edge_index, edge_attr, smiles=...)`` for the hot path (the graph part of the reference's
    ``from_smiles`` / ``from_InChI``, data/graph.py:12-64; ECFP, molar weight and ring counts are HabitchNN / PC-SAFT
    inputs and stay out of scope)."""
    import torch
    from .batching import Data
    g = smiles2graph(smiles)
    return Data(x=torch.from_numpy(g["node_feat"]), edge_index=torch.from_numpy(g["edge_index"]),
                edge_attr=torch.from_numpy(g["edge_feat"]), smiles=smiles, **labels)
