"""
The lineage-tree container of the PROSSTT API, restated for the MI355X path.

Mirrors ``prosstt.tree.Tree`` (reference: /root/reference/prosstt/tree.py:19-446):
same constructor, attributes and methods, so scripts written against the
reference keep working.  What is new sits behind the same names:

 * the (branch, time, gene) mean tensor lives on the device as one row-major
   binary32 matrix (rows = sum of branch lengths, branches in ``tree.branches``
   order); ``tree.means`` is a lazily materialised host view of it;
 * ``add_genes(relative_means, base)`` evaluates ``exp(rel) * base``
   (tree.py:181-182) with the ``means_from_rel`` HIP kernel.
"""
import itertools
from collections import defaultdict

import numpy as np
import pandas as pd

from . import device as _device

# every device mean tensor of the process gets its own number (Tree.means_token)
_DEVICE_TENSOR_SERIAL = itertools.count(1)


class Tree(object):
    """Formalization of a lineage tree (tree.py:19-80)."""

    def_time = 40
    def_genes = 500

    def __init__(self, topology=[["A", "B"], ["A", "C"]],
                 time={"A": def_time, "B": def_time, "C": def_time},
                 num_branches=3, branch_points=1, modules=None, G=def_genes,
                 density=None, root=None):
        self.topology = topology
        self.time = pd.Series(time, name="time")
        self.num_branches = num_branches
        self.branch_points = branch_points
        self.G = G
        self.branches = list(time.keys())
        self._host_means = None        # dict label -> (T_b, G) float64, or None
        self._dev_means = None         # torch float32 (sum T_b, G), or None
        self._dev_version = 0          # serial number of the device tensor this tree holds now (see means_token)
        self._dev_print = None         # fingerprint of _host_means when _dev_means was made from / mirrored to it
        self._lineage = None           # device cache left by simulate_lineage
        self._resident = None          # branches whose rows this process holds (None: all; see parallel.simulate_lineage_sharded)
        self._branch_owner = None      # branch -> rank when the tree is sharded over processes
        if modules is None:
            # consumes one draw of the global stream, as tree.py:67-68 does
            self.modules = 5 * branch_points + np.random.randint(1, 20)
        else:
            self.modules = modules
        self.root = self.branches[0] if root is None else root
        self.density = self.default_density() if density is None else density
        self._check_topology()

    # ---- validation the reference leaves implicit -----------
    def _check_topology(self):
        """branch_times() needs parents listed before children (tree.py:395-398)
        and the BFS needs every branch reachable from the root (sim_utils.py:560-566)."""
        seen = {self.root}
        for pair in self.topology:
            parent, child = pair[0], pair[1]
            if parent not in seen:
                raise ValueError("topology must list a branch as a child (or be the root) before it is "
                                 "used as a parent: %r appears as parent first" % (parent,))
            seen.add(child)
        missing = [b for b in self.branches if b not in seen]
        if missing and len(self.topology) > 0:
            raise ValueError("branches %r are not reachable from the root %r" % (missing, self.root))

    # ---- topology generators (tree.py:82-136) ------------------------------------
    @staticmethod
    def gen_random_topology(branch_points, branch_names=None):
        """Random binary topology with 2*branch_points+1 branches; one
        ``np.random.choice`` per branch point (tree.py:96-113)."""
        total_branches = 2 * branch_points + 1
        if branch_names is None:
            branch_names = np.arange(total_branches)
        open_ends = [0]
        unused = list(range(total_branches - 1, 0, -1))
        pairs = []
        while unused:
            parent = np.random.choice(open_ends)
            first, second = unused.pop(), unused.pop()
            pairs.append([branch_names[parent], branch_names[first]])
            pairs.append([branch_names[parent], branch_names[second]])
            open_ends.extend((first, second))
            open_ends.remove(parent)
        return pairs

    @classmethod
    def from_newick(cls, newick_tree, modules=None, genes=def_genes, density=None):
        """Lineage tree from a Newick string (tree.py:115-126).  Uses the ``newick`` package
        the reference depends on when it is installed, the built-in reader otherwise."""
        try:
            import newick
        except ImportError:
            from . import _newick as newick
        from . import tree_utils as tu
        parsed = newick.loads(newick_tree)
        top, time, branches, br_points, root = tu.parse_newick(parsed, cls.def_time)
        return cls(top, time, branches, br_points, modules, genes, density, root)

    @classmethod
    def from_random_topology(cls, branch_points, time, modules, genes):
        """tree.py:128-136."""
        topology = Tree.gen_random_topology(branch_points, branch_names=list(time.keys()))
        num_branches = len(np.unique(topology))
        return cls(topology, time, num_branches, branch_points, modules, genes)

    # ---- density (tree.py:138-151, 216-264) --------------------------------------
    def default_density(self):
        """Uniform density over all sum(T_b) positions."""
        total_time = 0
        for branch_time in self.time.values:
            total_time += branch_time
        return {k: np.array([1. / total_time] * int(self.time[k])) for k in self.time.keys()}

    def set_density(self, density):
        if not len(density) == len(self.branches):
            raise ValueError("The number of arrays in density must be equal to the number of "
                             "branches in the topology")
        for b in density:
            if not len(density[b]) == self.time[b]:
                raise ValueError("Branch %s was expected to have a length %s and instead is %s"
                                 % (b, self.time[b], np.shape(density[b])))
        self.density = density

    def set_velocity(self, velocity):
        from . import tree_utils as tu
        if not len(velocity) == len(self.branches):
            raise ValueError("The number of arrays in velocity must be equal to the number of "
                             "branches in the topology")
        for b in velocity:
            if not len(velocity[b]) == self.time[b]:
                raise ValueError("Branch %s was expected to have a length %s and instead is %s"
                                 % (b, self.time[b], np.shape(velocity[b])))
        velocity = tu.sanitize_velocity(velocity)
        self.density = tu._density_from_velocity(velocity)

    # ---- the mean tensor -----------------------------------------------------------
    def resident_branches(self):
        """Branches whose rows of the mean tensor this process holds, in ``tree.branches`` order: all of
        them, unless the lineage was built sharded over the ranks of a process group
        (``parallel.simulate_lineage_sharded``: every rank keeps the branches it owns)."""
        return list(self.branches) if self._resident is None else list(self._resident)

    def row_offsets(self):
        """Row of the first time step of every resident branch inside the device tensor."""
        offsets, at = {}, 0
        for b in self.resident_branches():
            offsets[b] = at
            at += int(self.time[b])
        return offsets, at

    @property
    def means(self):
        """dict branch -> (T_b, G) float64 (tree.py:64, 213).  Materialised from the device tensor
        on first access when the means were computed there (the values are the binary32 ones the
        sampler uses).  The arrays are ordinary writable arrays, as in the reference: the sampler
        reads the device tensor, which ``device_means`` refreshes whenever this dict or the contents
        of its arrays have changed since the last upload (a fingerprint is compared)."""
        if self._host_means is None and self._dev_means is not None:
            host = self._dev_means.cpu().numpy().astype(np.float64)
            offsets, _ = self.row_offsets()
            self._host_means = {b: host[offsets[b]:offsets[b] + int(self.time[b])] for b in self.resident_branches()}
            self._dev_print = self._means_fingerprint()
        return self._host_means

    @means.setter
    def means(self, value):
        self._host_means = value
        self._dev_means = None
        self._dev_print = None
        self._resident = None          # a dict of means covers the whole tree
        self._branch_owner = None

    def _means_fingerprint(self):
        return _device.host_fingerprint([self._host_means.get(b) if hasattr(self._host_means, "get")
                                         else self._host_means[b] for b in self.resident_branches()])

    def device_means(self):
        """(sum T_b, G) float32 device tensor.  When the host dict exists (the caller assigned it,
        or read ``tree.means``), its fingerprint decides whether the device copy is current:
        ``tree.means[b] = array`` and in-place edits both lead to a fresh upload; the caller's
        arrays are never frozen or modified."""
        if self._host_means is not None:
            now = self._means_fingerprint()
            if self._dev_means is None or now != self._dev_print:
                ctx = _device.get_context()
                import torch
                stacked = np.concatenate([np.asarray(self._host_means[b], dtype=np.float64)
                                          for b in self.resident_branches()], axis=0)
                as32 = stacked.astype(np.float32)
                tiny = np.float32(1.17549435e-38)      # positive means stay positive in binary32 (see means_from_rel)
                as32[(stacked > 0) & (as32 < tiny)] = tiny
                self._dev_means = ctx.tensor(as32, torch.float32)     # a private copy goes up
                self._dev_version = next(_DEVICE_TENSOR_SERIAL)
                self._dev_print = now
        if self._dev_means is None:
            raise ValueError("the tree has no gene expression yet: call add_genes first")
        return self._dev_means

    def means_token(self):
        """Changes whenever the content of ``device_means()`` does (a new upload or a new ``add_genes``): what the
        sampler's domain check keys its cached per-row flags of the mean tensor on."""
        self.device_means()            # compares the host fingerprint, uploads if the caller edited tree.means
        # a process-wide serial number, never reused: (id(tree), per-tree count) can repeat when a tree is freed and the
        # next one lands on the same address with its tensor on the same device address
        return ("means", self._dev_version)

    def add_genes(self, *args):
        """tree.py:154-163: one dict of average expression, or (relative means, base array)."""
        if len(args) == 1 and isinstance(args[0], dict):
            self._add_genes_from_average(args[0])
        if len(args) == 2 and isinstance(args[1], np.ndarray):
            self._add_genes_from_relative(args[0], args[1])

    def _add_genes_from_relative(self, relative_means, base_gene_expr):
        """means[b] = exp(relative_means[b]) * base  (tree.py:166-183), on the device."""
        import torch
        from . import simulation as sim
        ctx = _device.get_context()
        if len(relative_means) != len(self.resident_branches()):
            raise ValueError("The number of arrays in average_expression must be equal to the "
                             "number of branches in the topology")
        for b in self.resident_branches():
            shape = np.shape(relative_means[b])
            if shape != (self.time[b], self.G):
                raise ValueError("Branch %s was expected to have a shape %s and instead is %s"
                                 % (b, (self.time[b], self.G), shape))
        if np.shape(base_gene_expr) != (self.G,):
            raise ValueError("base_gene_expr must have one entry per gene")
        rel = sim._device_rel(self, relative_means)
        base = ctx.tensor(np.asarray(base_gene_expr, dtype=np.float64), torch.float64)
        self._dev_means = ctx.means_from_rel(rel, base)
        self._dev_version = next(_DEVICE_TENSOR_SERIAL)
        self._host_means = None
        self._dev_print = None

    def _add_genes_from_average(self, average_expression):
        """tree.py:186-213, with the reference's shape checks."""
        if not len(average_expression) == self.num_branches:
            raise ValueError("The number of arrays in average_expression must be equal to the "
                             "number of branches in the topology")
        for branch in average_expression:
            mean = average_expression[branch]
            if not mean.shape == (self.time[branch], self.G):
                raise ValueError("Branch %s was expected to have a shape %s and instead is %s"
                                 % (branch, (self.time[branch], self.G), mean.shape))
        self.means = average_expression

    # ---- pseudotime bookkeeping (tree.py:267-434) ----------------------------------
    def as_dictionary(self):
        treedict = defaultdict(list)
        for parent, child in self.topology:
            treedict[parent].append(child)
        return treedict

    def paths(self, start):
        """All root-to-leaf paths below ``start`` (tree.py:302-330)."""
        children = self.as_dictionary()

        def walk(node):
            if not children[node]:
                return [[node]]
            return [[node] + rest for kid in children[node] for rest in walk(kid)]
        return walk(start)

    def get_max_time(self):
        longest = 0
        for path in self.paths(self.root):
            longest = max(longest, np.sum([self.time[b] for b in path]))
        return int(longest)

    def morph_stack(self, stack):
        """Cumulative [start, end) of the branch lengths along one path (tree.py:402-423)."""
        begin = 0
        for i, length in enumerate(stack):
            stack[i] = [begin, begin + length]
            begin += length
        return stack

    def populate_timezone(self):
        """Pseudotime intervals that no branch boundary crosses (tree.py:332-374)."""
        zones = []
        stacks = [self.morph_stack(self.time[path].tolist()) for path in self.paths(self.root)]
        while stacks:
            starts = np.array([s[0][0] for s in stacks])
            ends = np.array([s[0][1] for s in stacks])
            if all(ends == np.max(ends)):
                zones.append([np.max(starts), np.max(ends) - 1])
                for s in stacks:
                    s.pop(0)
            else:
                cut = np.min(ends)
                zones.append([np.max(starts), cut - 1])
                for s in stacks:
                    if s[0][1] != cut:
                        s.insert(1, [cut, s[0][1]])
                    s.pop(0)
            stacks = [s for s in stacks if s]
        return zones

    def branch_times(self):
        """[first, last] absolute pseudotime of every branch (tree.py:376-399)."""
        branch_time = defaultdict(list)
        branch_time[self.root] = [0, self.time[self.root] - 1]
        for parent, child in self.topology:
            parent_end = branch_time[parent][1]
            branch_time[child] = [parent_end + 1, parent_end + self.time[child]]
        return branch_time

    def get_parallel_branches(self):
        """parent -> array of its children (tree.py:425-434)."""
        top_array = np.array(self.topology)
        return {b: top_array[top_array[:, 0] == b, 1] for b in np.unique(top_array[:, 0])}

    def default_gene_expression(self):
        """tree.py:436-446: lineage with a=0.05, base expression, means."""
        from . import simulation as sim
        from . import sim_utils as sut
        relative_expr, _, _ = sim.simulate_lineage(self, a=0.05)
        gene_scale = sut.simulate_base_gene_exp(self, relative_expr)
        self._add_genes_from_relative(relative_expr, gene_scale)
