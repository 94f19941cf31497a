"""
Utilities of the Tree class, mirroring ``prosstt.tree_utils``
(reference: /root/reference/prosstt/tree_utils.py): Newick parsing, text output
(the formats MERLoT-style benchmarks read) and the velocity -> density map.
Host-side I/O; nothing here is on the device path.
"""
import numpy as np
import pandas as pd


def parse_newick(newick_tree, def_time):
    """(topology, time, #branches, #branch points, root) of a parsed Newick tree
    (tree_utils.py:10-56).  Branch lengths of 0 become ``def_time``."""
    topology, time = [], {}
    branches = branch_points = 0
    root = None
    for node in newick_tree[0].walk():
        branches += 1
        time[node.name] = def_time if node.length == 0 else int(node.length)
        if not node.descendants:
            continue
        branch_points += 1
        topology.extend([node.name, child.name] for child in node.descendants)
        if node.ancestor is None:
            root = node.name
    return topology, time, branches, branch_points, root


def _labels(prefix, count):
    return ["%s_%d" % (prefix, i) for i in range(count)]


def _write_table(path, columns, index):
    """Tab-separated table with a header row and an index column (pandas' to_csv layout,
    which is what the reference's readers expect)."""
    pd.DataFrame(columns, index=index, columns=list(columns)).to_csv(path, sep="\t")


def save_cell_params(job_id, save_dir, labs, brns, scalings):
    """<save_dir>/<job_id>_cellparams.txt: pseudotime, branch and library size of every cell
    (tree_utils.py:59-83)."""
    _write_table("%s/%s_cellparams.txt" % (save_dir, job_id),
                 {"pseudotime": labs, "branches": brns, "scalings": scalings}, _labels("cell", len(labs)))


def save_gene_params(job_id, save_dir, gene_scale, alpha, beta):
    """<save_dir>/<job_id>_geneparams.txt: alpha, beta and base expression of every gene
    (tree_utils.py:86-110)."""
    _write_table("%s/%s_geneparams.txt" % (save_dir, job_id),
                 {"alpha": alpha, "beta": beta, "genescale": gene_scale}, _labels("gene", len(alpha)))


def save_matrices(job_id, save_dir, X, uMs, H):
    """Count matrix (<job>_simulation.txt, integers, tab-separated with cell/gene labels),
    coefficients (<job>_h.txt) and relative means per branch (<job>_ums<branch>.txt), the
    last two in numpy's savetxt format (tree_utils.py:113-145)."""
    stem = "%s/%s" % (save_dir, job_id)
    counts = pd.DataFrame(np.asarray(X), index=_labels("cell", X.shape[0]), columns=_labels("gene", X.shape[1]))
    counts.astype(int).to_csv(stem + "_simulation.txt", sep="\t")
    np.savetxt(stem + "_h.txt", H)
    for branch in uMs.keys():
        np.savetxt("%s_ums%s.txt" % (stem, branch), uMs[branch])


def save_matrices_npz(job_id, save_dir, X, uMs=None, H=None, compressed=False):
    """Binary alternative to ``save_matrices`` for matrices the text format cannot carry (the 50 000 x
    20 000 headline matrix is 4 GB as int32 and several times that as text): one
    ``<save_dir>/<job_id>_simulation.npz`` holding ``X`` as int32 (an int32 device tensor, or the
    ``device.PresentedCounts`` that ``draw_counts(..., out="torch")`` returns -- its rows are put back in plan order
    inside the copy -- travels to the host as it is, never widened to int64), ``H`` and ``ums<branch>`` when given.
    Row i is cell_i, column j gene_j, as in the text file.

    A ``scipy.sparse`` matrix (what ``draw_counts(..., out="csr")`` returns) is stored as it is, in the layout of
    ``scipy.sparse.save_npz`` (``data`` / ``indices`` / ``indptr`` / ``shape`` / ``format`` instead of ``X``):
    ``scipy.sparse.load_npz(path)`` gives the matrix back, ``np.load(path)`` the other arrays beside it."""
    if hasattr(X, "cell_of_row") and hasattr(X, "to_host"):      # device.PresentedCounts
        X = X.to_host("numpy32")
    if hasattr(X, "tocsr") and hasattr(X, "nnz"):                # scipy.sparse
        X = X.tocsr()
        if X.data.dtype != np.int32:
            if X.nnz and (X.data.min() < 0 or X.data.max() > np.iinfo(np.int32).max):
                raise ValueError("counts do not fit int32")
            X = X.astype(np.int32)
        arrays = {"format": np.array("csr".encode("ascii")), "shape": np.array(X.shape, dtype=np.int64), "data": X.data,
                  "indices": X.indices, "indptr": X.indptr}
        return _write_npz(job_id, save_dir, arrays, uMs, H, compressed)
    if hasattr(X, "detach"):                         # torch tensor
        X = X.detach().cpu().numpy()
    X = np.asarray(X)
    if X.dtype != np.int32:
        if X.size and (X.min() < 0 or X.max() > np.iinfo(np.int32).max):
            raise ValueError("counts do not fit int32")
        X = X.astype(np.int32)
    return _write_npz(job_id, save_dir, {"X": X}, uMs, H, compressed)


def _write_npz(job_id, save_dir, arrays, uMs, H, compressed):
    if H is not None:
        arrays["H"] = np.asarray(H)
    for branch in (uMs.keys() if uMs is not None else ()):
        arrays["ums%s" % branch] = np.asarray(uMs[branch])
    path = "%s/%s_simulation.npz" % (save_dir, job_id)
    (np.savez_compressed if compressed else np.savez)(path, **arrays)
    return path


def save_params(job_id, save_dir, lineage_tree, rseed):
    """<save_dir>/<job_id>_params.txt: genes, branch lengths, topology, #programs, seed
    (tree_utils.py:148-173)."""
    fields = [("Genes", lineage_tree.G), ("pseudotimes", list(lineage_tree.time.values)),
              ("topology", lineage_tree.topology), ("#modules", lineage_tree.modules), ("random seed", rseed)]
    with open("%s/%s_params.txt" % (save_dir, job_id), "w") as out:
        out.write("\n".join("%s: %s" % kv for kv in fields))


def sanitize_velocity(velocity, minimum_velocity=0.1):
    """Shift velocities so that all are positive (tree_utils.py:176-204)."""
    lowest = min([0] + [np.min(velocity[key]) for key in velocity])
    if lowest >= 0:
        return velocity
    for key in velocity:
        velocity[key] = velocity[key] + np.abs(lowest) + minimum_velocity
    return velocity


def _density_from_velocity(velocity):
    """Cell density along the tree from pseudotime velocities: where cells move fast, few are
    seen (tree_utils.py:207-242).  Over all branches together the density is the velocity
    mirrored inside its own range and normalised to sum to one over the tree.

    Bit for bit the reference's numbers (``np.random.choice`` builds its cdf from them, so the cells of a density
    plan hang on the last bit): the velocities are first divided by their total -- in place, the caller's arrays
    are normalised as the reference normalises them --, mirrored as ``(-v + max) + min`` in that order, and both
    totals are accumulated branch by branch (fixture g11, compared exactly)."""
    keys = list(velocity)
    flat = np.concatenate([np.asarray(velocity[k], dtype=float) for k in keys])
    cuts = np.cumsum([len(velocity[k]) for k in keys])[:-1]

    def branchwise_total(values):
        total = 0
        for piece in np.split(values, cuts):
            total += np.sum(piece)
        return total

    speed = branchwise_total(flat)
    scaled = flat / speed
    mirrored = (-scaled + flat.max() / speed) + flat.min() / speed
    mirrored = mirrored / branchwise_total(mirrored)
    for k, piece in zip(keys, np.split(scaled, cuts)):
        velocity[k] = piece
    return dict(zip(keys, np.split(mirrored, cuts)))
