"""FastaReader (reference src/util.rs:88-149): header verbatim, sequence lines concatenated."""


def read_fasta(path):
    recs, name, buf = [], None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(buf)))
                name, buf = line[1:], []
            elif name is not None:
                buf.append(line)
    if name is not None:
        recs.append((name, "".join(buf)))
    return recs
