"""Parser for the reference's per-region debug dump (`aux/inspect_<contig>.txt`).

The dump is what Contig::generate_inspect_file (src/Contig.cpp:368-453) + the Window printer
(src/Window.cpp:63-84) write when the three commented-out lines src/Hypo.cpp:262,265,271 are enabled
in a scratch copy of the reference.  One record per region:
    ==========(beg-end)\\t<TYPE>\\t<n_internal>\\t<n_pre>\\t<n_suf>\\t<n_empty>
    ++\\t<draft>
    ++\\t<consensus>
    <arms: internal, prefix, suffix in insertion order>
Only used by make_golden.py (in the build container) to harvest real-pipeline window INPUTS.
"""
from hypo_amd.batch import TextWindow

NON_WINDOW = ("SR", "MSR")


def parse(path, limit=None):
    """Yields (header, type, TextWindow, reference_consensus) for every window record."""
    with open(path) as f:
        f.readline()
        f.readline()
        line = f.readline()
        n = 0
        while line:
            assert line.startswith("=========="), line[:40]
            parts = line.rstrip("\n").split("\t")
            hdr, typ = parts[0], parts[1]
            ni, np_, ns, ne = (int(x) for x in parts[2:6])
            draft = f.readline().rstrip("\n")[3:]
            cons = f.readline().rstrip("\n")[3:]
            arms = []
            line = f.readline()
            while line and not line.startswith("=========="):
                arms.append(line.rstrip("\n"))
                line = f.readline()
            if typ in NON_WINDOW or ni + np_ + ns + ne == 0:
                continue
            assert len(arms) == ni + np_ + ns, (hdr, len(arms), ni, np_, ns)
            w = TextWindow(draft, arms[:ni], arms[ni:ni + np_], arms[ni + np_:], ne, typ == "LNG")
            yield hdr, typ, w, cons
            n += 1
            if limit and n >= limit:
                return
