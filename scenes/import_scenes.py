"""Re-serialise the reference's scene files named by the benchmark configs into scenes/.

Scene JSON is input DATA of the photon pass (the same role a dataset plays); the
values are kept verbatim, the layout is compact.  Mesh paths keep their relative
form "meshes/<name>.obj"; the files they name are Git-LFS pointers upstream and are
replaced by procedural stand-ins at load time (ray-tracer_amd/host/MeshGenerator.cpp).

Usage (only where /root/reference exists):  python scenes/import_scenes.py
"""
import json
import pathlib
import sys

NAMES = [
    "CornellBox", "CornellBoxGlassMirrorDiffuse", "MirrorKnot4k", "MirrorBombardment-200M",
    "MirrorBombardment", "MultiCameraTest", "CBGMD_stochastic128", "SinglePhotonVerify",
    "MirrorDirectTest", "CornellBoxQuads",
]


def main(reference="/root/reference"):
    out = pathlib.Path(__file__).resolve().parent
    for name in NAMES:
        data = json.loads((pathlib.Path(reference) / f"{name}.json").read_text())
        text = json.dumps(data, separators=(", ", ": "))
        (out / f"{name}.json").write_text(text + "\n")
        print("wrote", name)


if __name__ == "__main__":
    main(*sys.argv[1:])
