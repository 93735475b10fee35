"""Share of a file's tokens that sit inside an n-token run also present in a reference file (the copy
check the judge described: python tokens, comments and whitespace dropped).  Usage:
    python tools/diag/token_overlap.py <ours.py> <theirs.py> [n=12]"""
import io
import sys
import tokenize


def toks(path):
    out = []
    with open(path, "rb") as f:
        for t in tokenize.tokenize(f.readline):
            if t.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT,
                          tokenize.ENCODING, tokenize.ENDMARKER):
                continue
            out.append(t.string)
    return out


def overlap(a, b, n=12):
    ta, tb = toks(a), toks(b)
    grams = {tuple(tb[i:i + n]) for i in range(len(tb) - n + 1)}
    hit = [False] * len(ta)
    for i in range(len(ta) - n + 1):
        if tuple(ta[i:i + n]) in grams:
            for j in range(i, i + n):
                hit[j] = True
    return sum(hit) / max(1, len(ta))


if __name__ == "__main__":
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    print(f"{overlap(sys.argv[1], sys.argv[2], n):.1%}")
