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


def save_cell_params(job_id, save_dir, labs, brns, scalings):
    """<save_dir>/<job_id>_cellparams.txt (tree_utils.py:59-83)."""
    names = ["cell_" + str(i) for i in range(len(labs))]
    frame = pd.DataFrame({"pseudotime": labs, "branches": brns, "scalings": scalings},
                         index=names, columns=["pseudotime", "branches", "scalings"])
    frame.to_csv(save_dir + "/" + job_id + "_cellparams.txt", sep="\t")


def save_gene_params(job_id, save_dir, gene_scale, alpha, beta):
    """<save_dir>/<job_id>_geneparams.txt (tree_utils.py:86-110)."""
    names = ["gene_" + str(i) for i in range(len(alpha))]
    frame = pd.DataFrame({"alpha": alpha, "beta": beta, "genescale": gene_scale},
                         index=names, columns=["alpha", "beta", "genescale"])
    frame.to_csv(save_dir + "/" + job_id + "_geneparams.txt", sep="\t")


def save_matrices(job_id, save_dir, X, uMs, H):
    """Count matrix, relative means per branch and coefficients as text
    (tree_utils.py:113-145): _simulation.txt, _ums<branch>.txt, _h.txt."""
    cells = ["cell_" + str(i) for i in range(X.shape[0])]
    genes = ["gene_" + str(i) for i in range(X.shape[1])]
    pd.DataFrame(X, columns=genes, index=cells).astype(int).to_csv(
        save_dir + "/" + job_id + "_simulation.txt", sep="\t")
    np.savetxt(fname=save_dir + "/" + job_id + "_h.txt", X=H)
    for branch in uMs.keys():
        np.savetxt(fname=save_dir + "/" + job_id + "_ums" + str(branch) + ".txt", X=uMs[branch])


def save_params(job_id, save_dir, lineage_tree, rseed):
    """<save_dir>/<job_id>_params.txt (tree_utils.py:148-173)."""
    with open(save_dir + "/" + job_id + "_params.txt", 'w') as out:
        out.write("Genes: " + str(lineage_tree.G) + "\n")
        out.write("pseudotimes: " + str(list(lineage_tree.time.values)) + "\n")
        out.write("topology: " + str(lineage_tree.topology) + "\n")
        out.write("#modules: " + str(lineage_tree.modules) + "\n")
        out.write("random seed: " + str(rseed))


def sanitize_velocity(velocity, minimum_velocity=0.1):
    """Shift velocities so that all are positive (tree_utils.py:176-204)."""
    lowest = min([0] + [np.min(velocity[key]) for key in velocity])
    if lowest >= 0:
        return velocity
    for key in velocity:
        velocity[key] = velocity[key] + np.abs(lowest) + minimum_velocity
    return velocity


def _density_from_velocity(velocity):
    """Density inversely related to velocity (tree_utils.py:207-242; the reference's
    ``np.Inf`` no longer exists in numpy 2 -- ``np.inf`` here)."""
    total_velocity = 0
    global_min, global_max = np.inf, -np.inf
    for b in velocity:
        total_velocity += np.sum(velocity[b])
        global_max = max(global_max, np.max(velocity[b]))
        global_min = min(global_min, np.min(velocity[b]))
    global_min /= total_velocity
    global_max /= total_velocity
    density, total_density = {}, 0
    for b in velocity:
        velocity[b] = velocity[b] / total_velocity
        density[b] = - velocity[b] + global_max + global_min
        total_density += np.sum(density[b])
    for b in velocity:
        density[b] /= total_density
    return density
