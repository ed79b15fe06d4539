#!/usr/bin/env python3
"""tools/sim/run.py <model> ... -- run one or more of the filter models of this directory (README.md says what each is)."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = sorted(f[:-3] for f in os.listdir(HERE) if f.endswith(".py") and f not in ("run.py", "base.py"))


def main():
    if len(sys.argv) < 2 or sys.argv[1] in ("-l", "--list"):
        print("models:", " ".join(MODELS))
        return
    for m in sys.argv[1:]:
        if m not in MODELS:
            raise SystemExit(f"unknown model {m}; one of {MODELS}")
        print(f"== {m}")
        runpy.run_path(os.path.join(HERE, m + ".py"), run_name="__main__")


if __name__ == "__main__":
    main()
