"""Golden text of the reference's model views (print / report / simplereport / histogram), made by the REAL reference
(oracle/_ref/ref_driver view ...) — run in the build container only:  python tests/golden/make_views.py

For each case the reference trains and writes a model (views/<case>.colibri.patternmodel, data) and prints its views
(views/<case>.<view>.txt). tests/test_views.py loads the same model file through this repo's C++ face and compares
the text: byte-identical for report / histogram, identical as a set of lines for print (row order is the hash map's).
Class files: hamlet.colibri.cls is the reference's own fixture (exp/hamlet.colibri.cls); synthetic.colibri.cls maps class i -> "w<i>".
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402

DRIVER = oracle.REF_DRIVER
VIEWS = os.path.join(HERE, "views")
CASES = [  # (case, corpus, mode, maxlength, extra, class file)
    ("hamlet.u", "hamlet.v2", "u", 5, [], "hamlet.colibri.cls"),
    ("hamlet.us", "hamlet.v2", "us", 5, [], "hamlet.colibri.cls"),
    ("hamlet.i", "hamlet.v2", "i", 5, [], "hamlet.colibri.cls"),
    ("hamlet.is", "hamlet.v2", "is", 5, [], "hamlet.colibri.cls"),
    ("zipf20k.us", "zipf20k", "us", 3, ["-y", "3"], "synthetic.colibri.cls"),
    ("zipf20k.is", "zipf20k", "is", 3, [], "synthetic.colibri.cls"),
]


def main():
    os.makedirs(VIEWS, exist_ok=True)
    with open(os.path.join(HERE, "synthetic.colibri.cls"), "w") as f:
        for c in range(6, 600):
            f.write(f"{c}\tw{c}\n")
    for case, corpus, mode, l, extra, cls in CASES:
        model = os.path.join(VIEWS, f"{case}.colibri.patternmodel")
        subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{corpus}.colibri.dat"), mode, str(l), "2", "-q", "-o", model] + extra, stdout=subprocess.DEVNULL)
        kind = "i" if mode.startswith("i") else "u"
        for view in ("print", "report", "simplereport", "histogram"):
            out = subprocess.run([DRIVER, "view", model, kind, view, os.path.join(HERE, cls)], check=True, capture_output=True).stdout
            with open(os.path.join(VIEWS, f"{case}.{view}.txt"), "wb") as f:
                f.write(out)
    print("views written:", len(os.listdir(VIEWS)), "files")


if __name__ == "__main__":
    main()
