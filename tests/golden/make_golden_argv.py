#!/usr/bin/env python3
"""Extracts the ViralQuasispecies argument lists the reference's workflows build:
scripts/pipeline_per_stage.py's five run_* functions and polyte.py's single call, as DATA: for every call site the
flag names and their printf templates ("--edge_threshold=%f").  Runs only in the build container (reads
/root/reference); the output tests/golden/pipeline_argv.json is what tests/test_cli_flags.py checks hc-edgecalc against.
"""
import json
import os
import re

REF = "/root/reference"
SITES = [("scripts/pipeline_per_stage.py", None), ("polyte.py", None)]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pipeline_argv.json")


def main():
    sites = []
    for rel, _ in SITES:
        lines = open(os.path.join(REF, rel)).read().split("\n")
        i = 0
        while i < len(lines):
            # a call site: a list literal starting with the viralquasispecies binary
            if re.search(r"\[\s*viralquasispecies\s*,", lines[i]):
                start = i + 1
                flags = []
                j = i
                while j < len(lines):
                    for m in re.finditer(r'"--([A-Za-z_0-9]+)=([^"]*)"|"--([A-Za-z_0-9]+)"\s*,\s*"([^"]*)"', lines[j]):
                        if m.group(1):
                            flags.append({"flag": m.group(1), "template": m.group(2), "separate": False})
                        else:  # the two-token form: "--singles", "%s/singles.fastq"
                            flags.append({"flag": m.group(3), "template": m.group(4), "separate": True})
                    if re.search(r"subprocess\.check_call\(shell_command\)", lines[j]) or (re.search(r"^\s*\]\)?\s*$", lines[j]) and "shell_command" not in lines[i]):
                        break
                    j += 1
                # enclosing function name
                fn = next((re.match(r"def (\w+)", lines[k]).group(1) for k in range(i, -1, -1) if re.match(r"def (\w+)", lines[k])), "?")
                sites.append({"file": rel, "function": fn, "lines": [start, j + 1], "flags": flags})
                i = j
            i += 1
    json.dump({"source": "flag names and printf templates of the workflows' ViralQuasispecies calls", "sites": sites}, open(OUT, "w"), indent=1)
    for s in sites:
        print(s["file"], s["function"], s["lines"], len(s["flags"]))


if __name__ == "__main__":
    main()
