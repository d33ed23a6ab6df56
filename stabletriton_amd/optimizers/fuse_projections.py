"""Graph-level GEMM consolidation (additions; cf. the reference's unused fused
Q/K/V projection kernel, kernels/attention_proj.py:53-155):

* `fuse_shared_input_linears`: plain linear_wrapper calls that read the same tensor
  (to_q/to_k/to_v of self-attention, to_k/to_v of cross-attention, unet_pt.py:122-132;
  the 17 resnet time_emb_proj of SiLU(temb), unet_pt.py:72-76) become one
  `linear_cat_wrapper` GEMM; consumers get column slices (the attention kernel takes
  strided q/k/v, the conv epilogue a row-bias slice).
* `split_context`: everything that depends only on `encoder_hidden_states`
  (the 77-token text context: 140 K/V projections, step-invariant - SURVEY.md 8a
  row L) moves into a separate context GraphModule, evaluated once per prompt.
"""
from __future__ import annotations

import operator
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import fx, nn

from .wrappers import (layer_norm_wrapper, linear_cat_wrapper, linear_geglu_wrapper, linear_wrapper,
                       ln_linear_wrapper)


def fuse_shared_input_linears(gm: fx.GraphModule) -> int:
    groups: Dict[Tuple[fx.Node, bool], List[fx.Node]] = {}
    for n in gm.graph.nodes:
        if n.op == "call_function" and n.target is linear_wrapper and n.args[2] is False and isinstance(n.args[0], fx.Node):
            lin = gm.get_submodule(n.args[1].target)
            if isinstance(lin, nn.Linear):
                groups.setdefault((n.args[0], lin.bias is None), []).append(n)
    fused = 0
    for (src, _), nodes in groups.items():
        if len(nodes) < 2:
            continue
        first = nodes[0]
        mods = [gm.get_submodule(n.args[1].target) for n in nodes]
        if len({m.in_features for m in mods}) != 1 or len({m.weight.dtype for m in mods}) != 1:
            continue
        with gm.graph.inserting_before(first):
            attrs = tuple(gm.graph.get_attr(n.args[1].target) for n in nodes)
            cat = gm.graph.call_function(linear_cat_wrapper, (src, attrs))
            off = 0
            for n, m in zip(nodes, mods):
                sl = gm.graph.call_function(operator.getitem, (cat, (Ellipsis, slice(off, off + m.out_features))))
                n.replace_all_uses_with(sl)
                off += m.out_features
        for n in nodes:
            gm.graph.erase_node(n)
        fused += 1
    if fused:
        gm.graph.eliminate_dead_code()
        gm.graph.lint()
        gm.recompile()
    return fused


def fuse_layernorm_into_linear(gm: fx.GraphModule) -> int:
    """layer_norm_wrapper whose input comes from a GEMM and whose only consumer is a plain /
    concatenated / GEGLU projection disappears: the producing GEMM also emits per-row (sum, sum of
    squares) partials of what it stores, gamma is folded into the consumer's weights and the
    consumer applies mean/rstd as a rank-1 correction in its epilogue (`ln_linear_wrapper`)."""
    from .wrappers import linear_residual_wrapper
    count = 0
    for ln in list(gm.graph.nodes):
        if not (ln.op == "call_function" and ln.target is layer_norm_wrapper) or len(ln.users) != 1:
            continue
        user = next(iter(ln.users))
        if user.op != "call_function" or user.args[0] is not ln:
            continue
        lnmod = gm.get_submodule(ln.args[1].target)
        if len(lnmod.normalized_shape) != 1:
            continue
        if user.target is linear_wrapper and user.args[2] is False and len(user.args) == 3:
            linears, geglu = (user.args[1],), False
        elif user.target is linear_cat_wrapper:
            linears, geglu = tuple(user.args[1]), False
        elif user.target is linear_geglu_wrapper:
            linears, geglu = (user.args[1],), True
        else:
            continue
        kdim = lnmod.normalized_shape[0]
        mods = [gm.get_submodule(a.target) for a in linears]
        if any(m.in_features != kdim for m in mods) or kdim % 64 != 0:
            continue
        # the producer of the LayerNorm input must be one of our GEMMs (it will emit the partials)
        prod = ln.args[0]
        if not (isinstance(prod, fx.Node) and prod.op == "call_function"):
            continue
        if prod.target is linear_wrapper and prod.args[2] is False and len(prod.args) == 3:
            pass
        elif prod.target is linear_residual_wrapper and len(prod.args) == 3:
            pass
        else:
            continue
        pmod = gm.get_submodule(prod.args[1].target)
        if pmod.in_features % 64 != 0 or pmod.out_features != kdim:
            continue
        prod.args = tuple(prod.args) + (True,)            # emit_stats
        with gm.graph.inserting_after(prod):
            st = gm.graph.call_function(operator.getitem, (prod, 1))
            x0 = gm.graph.call_function(operator.getitem, (prod, 0))
        prod.replace_all_uses_with(x0, delete_user_cb=lambda u: u is not x0 and u is not st)
        with gm.graph.inserting_before(user):
            new = gm.graph.call_function(ln_linear_wrapper, (x0, st, ln.args[1], linears, geglu))
        user.replace_all_uses_with(new)
        gm.graph.erase_node(user)
        gm.graph.erase_node(ln)
        count += 1
    if count:
        gm.graph.eliminate_dead_code()
        gm.graph.lint()
        gm.recompile()
    return count


def fuse_query_projection_into_attention(gm: fx.GraphModule) -> int:
    """ln_linear_wrapper with ONE projection whose only consumer is the `q` of an attention_wrapper (the cross-attention
    query path: self-attention takes its q from the fused q|k|v GEMM through a slice) becomes
    `ln_linear_attention_wrapper`: the projection GEMM runs the attention core as its epilogue."""
    from .wrappers import attention_wrapper, ln_linear_attention_wrapper
    count = 0
    for att in list(gm.graph.nodes):
        if not (att.op == "call_function" and att.target is attention_wrapper) or att.kwargs:
            continue
        qn = att.args[0]
        if not (isinstance(qn, fx.Node) and qn.op == "call_function" and qn.target is ln_linear_wrapper and len(qn.users) == 1):
            continue
        x0, st, ln, linears = qn.args[0], qn.args[1], qn.args[2], qn.args[3]
        geglu = qn.args[4] if len(qn.args) > 4 else qn.kwargs.get("geglu", False)
        if geglu or len(linears) != 1 or att.args[1] is qn or att.args[2] is qn:
            continue
        with gm.graph.inserting_before(att):
            new = gm.graph.call_function(ln_linear_attention_wrapper,
                                         (x0, st, ln, linears[0], att.args[1], att.args[2], att.args[4], att.args[5], att.args[6]))
        att.replace_all_uses_with(new)
        gm.graph.erase_node(att)
        gm.graph.erase_node(qn)
        count += 1
    if count:
        gm.graph.eliminate_dead_code()
        gm.graph.lint()
        gm.recompile()
    return count


def split_context(gm: fx.GraphModule, context_arg: str = "encoder_hidden_states") -> Optional[fx.GraphModule]:
    """Move the sub-graph that depends only on `context_arg` into its own GraphModule.

    Returns the context module (ctx(encoder_hidden_states) -> tuple of tensors) or None if
    nothing could be hoisted.  `gm` gains a placeholder `context_cache` right after
    `context_arg`; its original placeholder stays (unused) so positions do not shift.
    """
    return split_region(gm, (context_arg,), "context_cache", class_name="ContextModule")


def split_region(gm: fx.GraphModule, sources: Sequence[str], cache_name: str, meta_sources: Sequence[str] = (),
                 stop_at_slices: bool = False, class_name: str = "RegionModule") -> Optional[fx.GraphModule]:
    """Move every node that depends only on the placeholders `sources` (plus parameters) into a new
    GraphModule `region(*meta_sources, *sources) -> tuple`; `gm` gains the placeholder `cache_name`
    after the last source and reads the region's outputs from it.

    `meta_sources` are placeholders the region may use for metadata only (`x.shape`, `x.dtype`,
    `x.device`): such getattr nodes (and pure functions of them) are duplicated into the region, not
    moved.  With `stop_at_slices` tensor slicing (`t[..., a:b]`) of a region tensor stays in `gm`,
    so one wide tensor crosses the boundary instead of its many views."""
    nodes = list(gm.graph.nodes)
    phs = {n.target: n for n in nodes if n.op == "placeholder"}
    if any(s_ not in phs for s_ in sources) or any(m not in phs for m in meta_sources):
        return None
    src_phs = [phs[s_] for s_ in sources]
    meta_phs = {phs[m] for m in meta_sources}
    inside = set(src_phs)              # depend on a source
    meta_only = set()                  # depend on metadata of a meta source (and constants) only
    region: List[fx.Node] = []
    for n in nodes:
        if n.op in ("placeholder", "output", "get_attr"):
            continue
        if (n.op == "call_function" and n.target is getattr and n.args[0] in meta_phs
                and n.args[1] in ("shape", "dtype", "device")):
            meta_only.add(n)
            continue
        ins = n.all_input_nodes
        if not ins:
            continue
        ok = all((i in inside) or (i in meta_only) or i.op == "get_attr" for i in ins)
        if not ok:
            continue
        if not any(i in inside for i in ins):
            if all(i in meta_only for i in ins):
                meta_only.add(n)
            continue
        if (stop_at_slices and n.op == "call_function" and n.target is operator.getitem
                and n.args[0].op != "placeholder" and isinstance(n.args[1], (tuple, slice))):
            continue
        inside.add(n)
        region.append(n)
    if not region:
        return None
    region_set = set(region)
    frontier = [n for n in region if any(u not in region_set for u in n.users)]
    if not frontier:
        return None

    # ---- build the region graph --------------------------------------------------------------
    cg = fx.Graph()
    env: Dict[fx.Node, fx.Node] = {}
    order = [n for n in nodes if n.op == "placeholder" and (n in meta_phs or n in src_phs)]
    for ph in order:
        env[ph] = cg.placeholder(ph.target)

    def remap(x):
        if x in env:
            return env[x]
        if x.op == "get_attr":
            env[x] = cg.get_attr(x.target)
        else:
            assert x in meta_only, f"unexpected external dependency {x.format_node()}"
            env[x] = cg.node_copy(x, remap)               # metadata helpers are duplicated on demand
        return env[x]

    for n in region:
        env[n] = cg.node_copy(n, remap)
    cg.output(tuple(env[n] for n in frontier))
    region_module = fx.GraphModule(gm, cg, class_name=class_name)
    region_module.input_names = [ph.target for ph in order]

    # ---- rewire the main graph ---------------------------------------------------------------
    last_src = [n for n in nodes if n.op == "placeholder" and n in src_phs][-1]
    with gm.graph.inserting_after(last_src):
        cache = gm.graph.placeholder(cache_name)
    first_user = next(n for n in gm.graph.nodes if n.op not in ("placeholder", "get_attr"))
    with gm.graph.inserting_before(first_user):
        for i, n in enumerate(frontier):
            gi = gm.graph.call_function(operator.getitem, (cache, i))
            n.replace_all_uses_with(gi)
    for n in reversed(region):
        if len(n.users) == 0:
            gm.graph.erase_node(n)
    gm.graph.eliminate_dead_code()
    gm.graph.lint()
    gm.recompile()
    return region_module
