"""Registers, scratch and waves per SIMD of the kernels the build compiled (from the compiler's remarks).

    python tools/kernel_resources.py [name-filter]
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "traceKernel"
    text = open(os.path.join(REPO, "ray-tracer_amd", "csrc", "build", "rt_kernels.resources")).read()
    for block in re.split(r"remark: Function Name: ", text)[1:]:
        name = block.split()[0]
        if want not in name:
            continue
        fields = {k.strip(): v for k, v in re.findall(r"remark:\s+([A-Za-z /\[\]]+?): (\w+) \[", block)}
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            short = re.search(r"(\w+(<[^>]*>)?)\(", name)
            name = short.group(1) if short else name
        except OSError:
            pass
        print(f"{name[-60:]:60s} vgpr {str(fields.get('VGPRs')):>4s} agpr {str(fields.get('AGPRs')):>3s} "
              f"scratch {str(fields.get('ScratchSize [bytes/lane]')):>4s} waves/SIMD {str(fields.get('Occupancy [waves/SIMD]'))} "
              f"lds {str(fields.get('LDS Size [bytes/block]'))}")


if __name__ == "__main__":
    main()
