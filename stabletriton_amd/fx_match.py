"""Module-aware sub-graph matching and replacement on torch.fx graphs.

Behavioural counterpart of the reference's matcher utility
(src/stabletriton/optimizers/utils/util.py:56-276 `SubgraphMatcher`,
util.py:344-524 `replace_pattern`, utils/fx.py:21-39 static-argument
equality), re-implemented from its behaviour:

* a pattern is any callable / nn.Module that fx can trace; its placeholders are
  wildcards and may bind either a graph node or a literal argument (so
  `num_heads`, `sm_scale` in the attention pattern bind the ints/floats the
  traced model carries);
* `call_module` nodes match by module *type* (plus an optional predicate), and
  the replacement is handed the *original* module so it reads
  weight/bias/eps from the live object at call time;
* non-node arguments must be equal; interior nodes of a match may not be used
  outside it; matches may not overlap (first match in graph order wins).

Unlike the reference (which copies a traced replacement graph and remaps module
targets), the replacement here is a builder callback that emits nodes straight
into the graph, or a traceable callable for reference-style use.
"""
from __future__ import annotations

import operator
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Union

import torch
from torch import fx, nn


@dataclass
class Match:
    anchor: fx.Node                                         # graph node producing the matched value
    nodes: Dict[fx.Node, fx.Node] = field(default_factory=dict)        # pattern node -> graph node
    bindings: Dict[str, Any] = field(default_factory=dict)             # placeholder name -> node | literal
    modules: Dict[str, str] = field(default_factory=dict)              # pattern module target -> graph target

    def interior(self) -> List[fx.Node]:
        return [g for p, g in self.nodes.items() if p.op != "placeholder"]


def _trace(pattern: Union[Callable, nn.Module, fx.GraphModule]) -> fx.GraphModule:
    return pattern if isinstance(pattern, fx.GraphModule) else fx.symbolic_trace(pattern)


class SubgraphMatcher:
    def __init__(self, pattern: Union[Callable, nn.Module], module_filter: Optional[Callable[[str, nn.Module], bool]] = None):
        self.pm = _trace(pattern)
        self.pgraph = self.pm.graph
        out = [n for n in self.pgraph.nodes if n.op == "output"][0]
        rets = out.all_input_nodes
        if len(rets) != 1:
            raise ValueError("patterns must return exactly one value")
        self.p_anchor = rets[0]
        self.module_filter = module_filter
        self.p_modules = dict(self.pm.named_modules())

    # ---- node-level comparison --------------------------------------------------------------
    def _same_target(self, pn: fx.Node, gn: fx.Node, g_modules, m: Match) -> bool:
        if pn.op != gn.op:
            return False
        if pn.op == "call_module":
            pmod, gmod = self.p_modules[pn.target], g_modules.get(gn.target)
            if gmod is None or type(pmod) is not type(gmod):
                return False
            if self.module_filter is not None and not self.module_filter(pn.target, gmod):
                return False
            prev = m.modules.get(pn.target)
            if prev is not None and prev != gn.target:
                return False
            return True
        return pn.target == gn.target

    def _match_arg(self, pa, ga, g_modules, m: Match) -> bool:
        if isinstance(pa, fx.Node):
            if pa.op == "placeholder":
                if pa.name in m.bindings:
                    prev = m.bindings[pa.name]
                    return prev is ga if isinstance(prev, fx.Node) or isinstance(ga, fx.Node) else prev == ga
                m.bindings[pa.name] = ga
                if isinstance(ga, fx.Node):
                    m.nodes[pa] = ga
                return True
            return isinstance(ga, fx.Node) and self._match_node(pa, ga, g_modules, m)
        if isinstance(ga, fx.Node):
            return False
        if isinstance(pa, (tuple, list)):
            return (isinstance(ga, (tuple, list)) and len(pa) == len(ga)
                    and all(self._match_arg(x, y, g_modules, m) for x, y in zip(pa, ga)))
        if isinstance(pa, dict):
            return (isinstance(ga, dict) and pa.keys() == ga.keys()
                    and all(self._match_arg(pa[k], ga[k], g_modules, m) for k in pa))
        if isinstance(pa, slice):
            return isinstance(ga, slice) and all(
                self._match_arg(getattr(pa, f), getattr(ga, f), g_modules, m) for f in ("start", "stop", "step"))
        return type(pa) is type(ga) and pa == ga

    def _match_node(self, pn: fx.Node, gn: fx.Node, g_modules, m: Match) -> bool:
        if pn in m.nodes:
            return m.nodes[pn] is gn
        if gn in m.nodes.values():
            return False                       # one graph node cannot play two pattern roles
        if not self._same_target(pn, gn, g_modules, m):
            return False
        if len(pn.args) != len(gn.args) or pn.kwargs.keys() != gn.kwargs.keys():
            return False
        m.nodes[pn] = gn
        if pn.op == "call_module":
            m.modules[pn.target] = gn.target
        for pa, ga in zip(pn.args, gn.args):
            if not self._match_arg(pa, ga, g_modules, m):
                return False
        for k in pn.kwargs:
            if not self._match_arg(pn.kwargs[k], gn.kwargs[k], g_modules, m):
                return False
        return True

    # ---- whole-graph search -----------------------------------------------------------------
    def match(self, gm: fx.GraphModule) -> List[Match]:
        g_modules = dict(gm.named_modules())
        found: List[Match] = []
        taken = set()
        for gn in gm.graph.nodes:
            if gn.op in ("placeholder", "output"):
                continue
            m = Match(anchor=gn)
            if not self._match_node(self.p_anchor, gn, g_modules, m):
                continue
            interior = m.interior()
            inside = set(interior)
            # interior values (other than the anchor) must not leak out of the match
            if any(u not in inside for n in interior if n is not gn for u in n.users):
                continue
            if any(n in taken for n in interior):
                continue
            taken.update(interior)
            found.append(m)
        return found


def replace_pattern(gm: fx.GraphModule, pattern: Union[Callable, nn.Module], replacement: Callable,
                    module_filter: Optional[Callable[[str, nn.Module], bool]] = None) -> List[Match]:
    """Rewrite every match of `pattern` in `gm`.

    `replacement(graph, match) -> fx.Node` emits the new value; it runs with the
    insertion point set just before the matched anchor.  `match.bindings` holds
    what each pattern placeholder bound, `match.modules` maps the pattern's
    module attribute names to the live module targets in `gm`.
    """
    matches = SubgraphMatcher(pattern, module_filter).match(gm)
    order = {n: i for i, n in enumerate(gm.graph.nodes)}
    replaced: Dict[fx.Node, fx.Node] = {}
    for m in matches:
        # a wildcard of this match may have bound the anchor of an earlier match (chained patterns):
        # hand the replacement the node that now carries that value
        for name, val in list(m.bindings.items()):
            while isinstance(val, fx.Node) and val in replaced:
                val = replaced[val]
            m.bindings[name] = val
        with gm.graph.inserting_before(m.anchor):
            new = replacement(gm.graph, m)
        m.anchor.replace_all_uses_with(new)
        replaced[m.anchor] = new
        for n in sorted(m.interior(), key=order.__getitem__, reverse=True):
            if len(n.users) == 0:
                gm.graph.erase_node(n)
    if matches:
        gm.graph.lint()
        gm.recompile()
    return matches


def module_of(gm: fx.GraphModule, m: Match, pattern_attr: str) -> nn.Module:
    return gm.get_submodule(m.modules[pattern_attr])


def get_attr_node(graph: fx.Graph, target: str) -> fx.Node:
    """`self.<target>` as a graph value (used to pass live modules to wrappers)."""
    return graph.get_attr(target)


getitem = operator.getitem
