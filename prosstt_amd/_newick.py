"""
Minimal Newick reader used by ``Tree.from_newick`` when the third-party ``newick`` package
(a dependency of the reference, setup.py:12) is not installed.  It provides just what
``tree_utils.parse_newick`` reads: ``loads(text)[0].walk()`` in pre-order over nodes with
``name``, ``length`` (0.0 when absent), ``descendants`` and ``ancestor``.
"""


class Node:
    def __init__(self, name=None, length=0.0):
        self.name = name
        self.length = length
        self.descendants = []
        self.ancestor = None

    def add(self, child):
        child.ancestor = self
        self.descendants.append(child)

    def walk(self):
        yield self
        for child in self.descendants:
            yield from child.walk()


def _parse(text, pos):
    """subtree := [ '(' subtree {',' subtree} ')' ] [name] [':' length]"""
    node = Node()
    if pos < len(text) and text[pos] == "(":
        pos += 1
        while True:
            child, pos = _parse(text, pos)
            node.add(child)
            if pos >= len(text):
                raise ValueError("unbalanced parentheses in Newick string")
            if text[pos] == ",":
                pos += 1
                continue
            if text[pos] == ")":
                pos += 1
                break
            raise ValueError("unexpected %r at position %d of Newick string" % (text[pos], pos))
    start = pos
    while pos < len(text) and text[pos] not in ",():;":
        pos += 1
    label = text[start:pos].strip()
    node.name = label if label else None
    if pos < len(text) and text[pos] == ":":
        pos += 1
        start = pos
        while pos < len(text) and text[pos] not in ",();":
            pos += 1
        node.length = float(text[start:pos])
    return node, pos


def loads(text):
    """List of the trees in ``text`` (one per ';'-terminated statement)."""
    trees = []
    for statement in text.strip().split(";"):
        statement = "".join(statement.split())
        if not statement:
            continue
        node, pos = _parse(statement, 0)
        if pos != len(statement):
            raise ValueError("trailing characters in Newick string: %r" % statement[pos:])
        trees.append(node)
    return trees
