"""
Static layout of the message-passing levels: which irreps a level carries, in which order, and
which (source, r1, r2) blocks make up the channel axis of every CatMix input.

The reference never states this layout: it emerges from Python dict / set iteration order
(lgn/cg_lib/cg_ops.py:179-215, lgn/g_lib/g_torch.py:203-214, lgn/g_lib/parameter_dict_new.py:14-15,
lgn/models/lgn_levels.py:210,224).  Checkpoints written by the reference are only loadable if the
same layout is reproduced, so the bookkeeping below replays the same constructions (dict insertion
order + the ``set`` of parameter keys) and tests/test_host.py pins the outcome to the fixtures.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

Irrep = Tuple[int, int]


def param_key_order(keys_in_registration_order: Sequence[Irrep]) -> List[Irrep]:
    """Order in which a MixReps emits its parts: iteration order of
    ``set(map(eval, parameter_names))`` (parameter_dict_new.py:14-15, g_torch.py:238-241)."""
    names = [str(k) for k in keys_in_registration_order]
    out = []
    for item in set(map(_parse, names)):
        out.append(item)
    return out


def _parse(name: str) -> Irrep:
    a, b = name.strip("() ").split(",")
    return (int(a), int(b))


def product_tau(tau1: Dict[Irrep, int], tau2: Dict[Irrep, int], maxdim: int) -> Dict[Irrep, int]:
    """cg_product_tau (lgn/cg_lib/cg_ops_tau.py:6-44), including its dict insertion order."""
    out: Dict[Irrep, int] = {}
    for (k1, n1), t1 in tau1.items():
        for (k2, n2), t2 in tau2.items():
            if max(k1, n1, k2, n2) >= maxdim:
                continue
            for k in range(abs(k1 - k2), min(k1 + k2, maxdim - 1) + 1, 2):
                for n in range(abs(n1 - n2), min(n1 + n2, maxdim - 1) + 1, 2):
                    out[(k, n)] = out.get((k, n), 0) + t1 * t2
    return out


def cg_tau_out(tau1: Dict[Irrep, int], tau2: Dict[Irrep, int], maxdim: int) -> Dict[Irrep, int]:
    """CGProduct.tau_out (lgn/cg_lib/cg_ops.py:99-111): channels x number of contributing pairs."""
    ones1 = {k: int(v > 0) for k, v in tau1.items()}
    ones2 = {k: int(v > 0) for k, v in tau2.items()}
    chans = set([t for t in tau1.values() if t > 0] + [t for t in tau2.values() if t > 0])
    if len(chans) != 1:
        raise ValueError(f"CG products need the same number of channels in every part, got {tau1} x {tau2}")
    nchan = chans.pop()
    return {k: nchan * t for k, t in product_tau(ones1, ones2, maxdim).items()}


def cat_tau(taus: Sequence[Dict[Irrep, int]], maxdim: int) -> Dict[Irrep, int]:
    """CatReps.tau_out (lgn/nn/g_nn.py:150-158)."""
    out: Dict[Irrep, int] = {}
    for tau in taus:
        for key, val in tau.items():
            if val > 0 and max(key) <= maxdim - 1:
                out[key] = out.get(key, 0) + val
    return out


@dataclass
class LevelPlan:
    channels_in: int
    channels_out: int
    node_order: List[Irrep]                 # run-time order of the incoming node GVec
    tau_in: Dict[Irrep, int]                # bookkeeping dicts (insertion order matters)
    tau_edge: Dict[Irrep, int]
    tau_ag: Dict[Irrep, int]
    tau_sq: Dict[Irrep, int]
    tau_cat: Dict[Irrep, int]
    tau_out: Dict[Irrep, int]               # == CatMix parameter registration order
    out_order: List[Irrep] = field(default_factory=list)   # order of the level's output GVec
    # per output irrep: channel blocks of the CatMix input, in order: (source, r1, r2)
    cat_blocks: Dict[Irrep, List[Tuple[str, Irrep, Irrep]]] = field(default_factory=dict)


def _product_blocks(order1: Sequence[Irrep], order2: Sequence[Irrep], maxdim: int, tag: str):
    """Replay the pair loop of cg_product (cg_ops.py:163-215) on key orders only."""
    maxk1 = max(k for k, _ in order1); maxn1 = max(n for _, n in order1)
    maxk2 = max(k for k, _ in order2); maxn2 = max(n for _, n in order2)
    max_dim = min(max(maxk1 + maxk2, maxn1 + maxn2) + 1, maxdim)
    blocks: Dict[Irrep, List[Tuple[str, Irrep, Irrep]]] = {}
    for (k1, n1) in order1:
        for (k2, n2) in order2:
            if max(k1, n1, k2, n2) > max_dim - 1:
                continue
            for k in range(abs(k1 - k2), min(maxdim, k1 + k2 + 1), 2):
                for n in range(abs(n1 - n2), min(maxdim, n1 + n2 + 1), 2):
                    blocks.setdefault((k, n), []).append((tag, (k1, n1), (k2, n2)))
    return blocks


def build_level_plans(num_channels: Sequence[int], maxdim: Sequence[int], max_zf: Sequence[int], mlp: bool,
                      tau_node_in: Dict[Irrep, int], node_order_in: Sequence[Irrep]) -> List[LevelPlan]:
    """Bookkeeping of LGNCG.__init__ (lgn/models/lgn_cg.py:79-110) + LGNNodeLevel.__init__
    (lgn/models/lgn_levels.py:52-94) + the run-time key orders."""
    plans: List[LevelPlan] = []
    tau = dict(tau_node_in)
    order = list(node_order_in)
    for lvl in range(len(num_channels) - 1):
        md = maxdim[lvl]
        tau_edge = {(l, l): num_channels[lvl] for l in range(max_zf[lvl] + 1)}
        tau_sq = cg_tau_out(tau, tau, md)
        tau_ag = cg_tau_out(tau, tau_edge, md)
        tau_cat = cat_tau([tau_ag, tau, tau_sq], md)
        tau_out = {k: num_channels[lvl + 1] for k, v in tau_cat.items() if v}
        out_order = param_key_order(sorted(tau_out.keys()))   # ParameterDict registers a plain dict in sorted key order
        if mlp:   # CGMLP pops (0,0) and re-inserts it last (lgn_levels.py:210,224)
            out_order = [k for k in out_order if k != (0, 0)] + [(0, 0)]
        ag = _product_blocks(order, list(tau_edge.keys()), md, "ag")
        sq = _product_blocks(order, order, md, "sq")
        cat_blocks = {}
        for key in tau_cat:
            blocks = list(ag.get(key, []))
            if key in order:
                blocks.append(("node", key, key))
            blocks += sq.get(key, [])
            cat_blocks[key] = blocks
            assert len(blocks) * num_channels[lvl] == tau_cat[key], (key, blocks, tau_cat)
        plans.append(LevelPlan(num_channels[lvl], num_channels[lvl + 1], list(order), dict(tau), tau_edge, tau_ag,
                               tau_sq, tau_cat, tau_out, out_order, cat_blocks))
        tau = dict(tau_out)        # lgn_cg.py:105 hands the CatMix dict (not the GVec order) to the next level
        order = out_order
    return plans


# the only layout the maxdim=2 fused kernels implement (SURVEY 8 a-3'); asserted at construction time
MAXDIM2_BLOCKS = {
    (1, 1): [("ag", (1, 1), (0, 0)), ("ag", (0, 0), (1, 1)), ("node", (1, 1), (1, 1)),
             ("sq", (1, 1), (0, 0)), ("sq", (0, 0), (1, 1))],
    (0, 0): [("ag", (1, 1), (1, 1)), ("ag", (0, 0), (0, 0)), ("node", (0, 0), (0, 0)),
             ("sq", (1, 1), (1, 1)), ("sq", (0, 0), (0, 0))],
}


def check_maxdim2_layout(plan: LevelPlan):
    got = {k: plan.cat_blocks[k] for k in plan.cat_blocks}
    if got != MAXDIM2_BLOCKS or plan.node_order != [(1, 1), (0, 0)]:
        raise RuntimeError(f"unexpected maxdim=2 CatMix layout {got} / node order {plan.node_order}; "
                           "the fused kernels assume SURVEY 8 a-3'")


# ---------------------------------------------------------------------------------------------------
# sparse tables of the generic (any maxdim) level: which products feed which concatenated row
# ---------------------------------------------------------------------------------------------------
def irrep_dim(r: Irrep) -> int:
    return (r[0] + 1) * (r[1] + 1)


def packed_offsets(order: Sequence[Irrep]) -> Tuple[Dict[Irrep, int], int]:
    """Component offset of every irrep inside a packed feature vector with the irreps in ``order``; total length Q."""
    off, q = {}, 0
    for r in order:
        off[r] = q
        q += irrep_dim(r)
    return off, q


def build_local_tables(plan: LevelPlan, cg_dict, tol: float = 1e-14, weight_offsets: Dict[Irrep, int] = None):
    """Flatten (aggregate CG, power CG, concatenation) of one level into CSR term lists for
    csrc/generic_local.hip (struct lgn_local_tables of include/lgn_amd.h).

    Packed input components: irreps in ``plan.node_order``; packed output components: ``plan.out_order``.
    Row (l, block, m) of the concatenated CatMix input of output irrep l is a sparse combination of
      type 0: moments U[q][k]   (k = 0: edge irrep (0,0); k = 1 + m2: edge irrep (1,1))   -- aggregate blocks
      type 1: node component X[q]                                                             -- node block
      type 2: products X[q1] X[q2]                                                            -- power blocks
    with the Clebsch-Gordan coefficient as weight (reference: the CG matmul of cg_ops.py:195-204 applied to the
    Kronecker index m1*d2 + m2 of cg_ops.py:281-288).

    ``weight_offsets`` ({irrep: offset in doubles of its CatMix weight (2, CO, K) from the level's weight base}): where the
    kernels find each irrep's weights (and write its gradient partials).  Default: the weights concatenated in
    ``plan.out_order`` (what the per-operator autograd path passes); the whole-network native calls give the offsets
    inside the flat parameter block instead, so that nothing is gathered or scattered at run time."""
    import numpy as np
    qoff, q = {}, 0
    for r in plan.node_order:
        qoff[r] = q
        q += irrep_dim(r)
    Q = q
    out_irreps = list(plan.out_order)
    row_ptr, t_type, t_a, t_b, t_coef = [0], [], [], [], []
    out_dim, out_nblk, out_row0, out_q0, out_w0 = [], [], [], [], []
    qo, wbase, nrows = 0, 0, 0
    C, CO = plan.channels_in, plan.channels_out
    for L in out_irreps:
        d = irrep_dim(L)
        if d not in (1, 3, 4, 9):
            raise NotImplementedError(f"irrep {L} (dimension {d}): the table-driven kernels are instantiated for maxdim <= 3")
        blocks = plan.cat_blocks[L]
        out_dim.append(d); out_nblk.append(len(blocks)); out_row0.append(nrows); out_q0.append(qo)
        out_w0.append(2 * wbase if weight_offsets is None else int(weight_offsets[L]))
        qo += d
        wbase += CO * len(blocks) * C
        for (src, r1, r2) in blocks:
            d1, d2 = irrep_dim(r1), irrep_dim(r2)
            if src != "node":
                cgm = cg_dict[(r1, r2)][L].detach().cpu().numpy().astype(np.float64).reshape(d, d1, d2)
            for m in range(d):
                if src == "node":
                    t_type.append(1); t_a.append(qoff[L] + m); t_b.append(0); t_coef.append(1.0)
                else:
                    for m1 in range(d1):
                        for m2 in range(d2):
                            cf = float(cgm[m, m1, m2])
                            if abs(cf) <= tol:
                                continue
                            if src == "ag":
                                k = 0 if r2 == (0, 0) else 1 + m2
                                t_type.append(0); t_a.append((qoff[r1] + m1) * 5 + k); t_b.append(0)
                            else:
                                t_type.append(2); t_a.append(qoff[r1] + m1); t_b.append(qoff[r2] + m2)
                            t_coef.append(cf)
                if len(t_type) == row_ptr[-1]:      # a row whose CG coefficients all vanish still needs one (null) term:
                    t_type.append(1); t_a.append(0); t_b.append(0); t_coef.append(0.0)      # the walk marks row ends on terms
                t_type[-1] |= 4                     # bit 2: last term of its row (the type is t_type & 3)
                row_ptr.append(len(t_type))
                nrows += 1
    # transposed lists
    u_lists = [[] for _ in range(Q * 5)]
    x_lists = [[] for _ in range(Q)]
    for row in range(nrows):
        for t in range(row_ptr[row], row_ptr[row + 1]):
            if t_type[t] & 3 == 0:
                u_lists[t_a[t]].append((row, t_coef[t]))
            elif t_type[t] & 3 == 1:
                x_lists[t_a[t]].append((row, -1, t_coef[t]))
            else:
                x_lists[t_a[t]].append((row, t_b[t], t_coef[t]))
                x_lists[t_b[t]].append((row, t_a[t], t_coef[t]))
    u_ptr, u_row, u_coef = [0], [], []
    for lst in u_lists:
        for (row, cf) in lst:
            u_row.append(row); u_coef.append(cf)
        u_ptr.append(len(u_row))
    x_ptr, x_row, x_other, x_coef = [0], [], [], []
    for lst in x_lists:
        for (row, other, cf) in lst:
            x_row.append(row); x_other.append(other); x_coef.append(cf)
        x_ptr.append(len(x_row))
    ints = dict(row_ptr=row_ptr, t_type=t_type, t_a=t_a, t_b=t_b, out_dim=out_dim, out_nblk=out_nblk, out_row0=out_row0,
                out_q0=out_q0, out_w0=out_w0, u_ptr=u_ptr, u_row=u_row, x_ptr=x_ptr, x_row=x_row, x_other=x_other)
    dbls = dict(t_coef=t_coef, u_coef=u_coef, x_coef=x_coef)
    n_units = sum(((d + 3) // 4) * nb for d, nb in zip(out_dim, out_nblk))       # (irrep, chunk of <= 4 rows, block) units of the forward walk
    return dict(Q=Q, Qout=qo, n_rows=nrows, n_out=len(out_irreps), n_w=wbase, n_units=n_units, ints=ints, dbls=dbls,
                in_irreps=list(plan.node_order), out_irreps=out_irreps)


# ---------------------------------------------------------------------------------------------------
# the two level kinds of maxdim = 3 networks, matched against compile-time tables (csrc/cg_static_tables.hpp)
# ---------------------------------------------------------------------------------------------------
_STATIC_CACHE = {}


def canonical_static_tables():
    """{kind: tables} of the first level (kind 1: node irreps (1,1), (0,0)) and of a later level (kind 2: all five irreps) of a
    maxdim = 3 network with CGMLP levels; the channel counts do not enter the walk tables."""
    if not _STATIC_CACHE:
        from .cg_lib import CGDict
        cg = CGDict(maxdim=3)
        tau0 = {(0, 0): 2, (1, 1): 2}
        plans = build_level_plans([2, 2, 2], [3, 3], [1, 1], True, tau0, param_key_order(sorted(tau0)))
        for kind, plan in ((1, plans[0]), (2, plans[1])):
            _STATIC_CACHE[kind] = build_local_tables(plan, cg)
    return _STATIC_CACHE


_STATIC_KEYS_I = ("out_dim", "out_nblk", "out_row0", "out_q0", "row_ptr", "t_type", "t_a", "t_b", "u_ptr", "u_row")
_STATIC_KEYS_D = ("t_coef", "u_coef")


def static_kind(tab: dict) -> int:
    """1 / 2 when a level's tables are exactly those compiled into csrc/cg_static_tables.hpp, else 0."""
    for kind, ref in canonical_static_tables().items():
        if (tab["Q"] == ref["Q"] and tab["Qout"] == ref["Qout"] and tab["n_rows"] == ref["n_rows"]
                and all(list(tab["ints"][k]) == list(ref["ints"][k]) for k in _STATIC_KEYS_I)
                and all(list(tab["dbls"][k]) == list(ref["dbls"][k]) for k in _STATIC_KEYS_D)):
            return kind
    return 0
