"""Golden fixtures of the class encoder, made by the REAL reference (oracle/_ref/ref_driver encode = ClassEncoder::build /
save / encodefile driven like src/classencode.cpp:134-198) — run in the build container only:  python tests/golden/make_classenc_golden.py

Inputs: classenc/apology.txt (the reference's own fixture exp/apology.txt), classenc/quirks.txt and classenc/zipf.txt (written here).
Outputs per case: classenc/<case>.colibri.cls / .colibri.dat / .rc (exit status: 0 ok, 4 = unknown token in strict mode).
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402

G = os.path.join(HERE, "classenc")

QUIRKS = (b"the cat sat on the mat .\n"
          b"the  cat   sat\ton the\tmat . \n"            # runs of spaces, tabs inside and at the end of words
          b"\n"                                           # empty line
          b"windows line ends here\r\n"                   # \r is right-trimmed from the last word
          b"\r\n"                                         # a line that is only \r: skipped as a word, still a line
          b" leading space and trailing space \n"
          b"\tleadingtab stays\t\t\n"                     # trim is a RIGHT trim: the leading tab belongs to the word
          b"\t\r \t\t \r\r x\n"                           # single \t / \r segments are skipped, multi-byte ones become the empty word
          b"lone tab before the final space \t \n"   # "\t " is cut as one word by the frequency list, passes its filter and trims to ""
          b"lone cr before two final spaces \r  \n"
          b"back\bspace\b \bx\b\n"                        # \b is trimmed by the encoder but not by the frequency list
          b"gaps {*} and {**} and {?} and {*3*} and {*0*} here\n"
          b"{*x*} {* {**}x {|} }{\n"
          + "naïve café 日本語 \U0001F600 the\n".encode("utf-8")
          + b"long " + b"x" * 300 + b" word " + b"y" * 191 + b" " + b"z" * 192 + b"\n"
          b"the the the cat cat mat\n"
          b"last line has no newline and is not encoded")


def zipf_text(seed=5, nlines=400):
    rng = np.random.default_rng(seed)
    words = [f"w{i}" for i in range(300)] + ["ab", "a", "b", "ba", "aa"]
    p = 1.0 / np.arange(1, len(words) + 1)
    p /= p.sum()
    lines = []
    for _ in range(nlines):
        n = int(rng.integers(0, 25))
        lines.append(" ".join(words[i] for i in rng.choice(len(words), size=n, p=p)))
    return ("\n".join(lines) + "\n").encode()


CASES = [  # (case, text file, options for ref_classencode)
    ("apology", "apology.txt", {}),
    ("apology.t3U", "apology.txt", {"threshold": 3, "allowunknown": True}),
    ("apology.t3", "apology.txt", {"threshold": 3}),  # strict: unknown token -> status 4
    ("quirks", "quirks.txt", {}),                      # strict: "back\bspace\b" is counted with its \b but looked up without -> status 4
    ("quirks.U", "quirks.txt", {"allowunknown": True}),
    ("zipf", "zipf.txt", {}),                         # many equal frequencies: the tie order is libstdc++'s unordered_map order
    ("zipf.t2U", "zipf.txt", {"threshold": 2, "allowunknown": True}),
    ("quirks.c_apologyU", "quirks.txt", {"cls": "apology", "allowunknown": True}),
    ("quirks.c_apology", "quirks.txt", {"cls": "apology"}),
    ("quirks.c_apology.e", "quirks.txt", {"cls": "apology", "extend": True}),
    ("zipf.c_quirksU.e.t2", "zipf.txt", {"cls": "quirks.U", "extend": True, "threshold": 2}),
]


def main():
    os.makedirs(G, exist_ok=True)
    open(os.path.join(G, "quirks.txt"), "wb").write(QUIRKS)
    open(os.path.join(G, "zipf.txt"), "wb").write(zipf_text())
    for case, textfile, opt in CASES:
        kw = dict(opt)
        if "cls" in kw:
            kw["cls_path"] = os.path.join(G, kw.pop("cls") + ".colibri.cls")
        for ext in ("cls", "dat"):
            p = os.path.join(G, f"{case}.colibri.{ext}")
            if os.path.exists(p):
                os.remove(p)
        rc = oracle.ref_classencode(os.path.join(G, textfile), os.path.join(G, case), **kw)
        open(os.path.join(G, f"{case}.rc"), "w").write(str(rc))
        if rc != 0:  # the reference leaves a partial .dat behind when it throws; it is not part of the contract
            for ext in ("cls", "dat"):
                p = os.path.join(G, f"{case}.colibri.{ext}")
                if os.path.exists(p):
                    os.remove(p)
        print(case, "status", rc, [os.path.getsize(os.path.join(G, f"{case}.colibri.{e}")) if os.path.exists(os.path.join(G, f"{case}.colibri.{e}")) else None for e in ("cls", "dat")])


if __name__ == "__main__":
    main()
