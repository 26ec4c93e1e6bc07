"""Host-side scene loader: the reference's key set, ordering rules and error behaviour
(src/SceneLoader.cpp, src/MaterialLibrary.cpp, src/Library.cpp)."""
import json
import math
import os

import numpy as np
import pytest

from conftest import REPO


def write(tmp_path, data, name="scene.json"):
    p = tmp_path / name
    p.write_text(data if isinstance(data, str) else json.dumps(data))
    return str(p)


BASE = {
    "$renderConfiguration": {"$width": 8, "$height": 6, "$photonsPerLight": 10, "$bounceThreshold": 2},
    "$scene": {"Camera": {"$type": "Camera", "$verticalFieldOfView": 60.0}},
}


def test_defaults_and_settings(rt, tmp_path):
    scene = rt.Scene(write(tmp_path, BASE))
    s = scene.settings
    assert (s.image_width, s.image_height, s.photons_per_light, s.bounce_threshold) == (8, 6, 10, 2)
    # defaults of include/RenderSettings.h
    assert s.use_probe_gather == 1 and s.bounce_store_capacity == 40_000_000 and s.fetch_size == 100000
    assert s.termination_threshold == 1.0 and s.probe_keep_radius_scale == 1.5 and s.splat_min_radius_scale == 0.5
    assert s.seeded == 0 and s.deterministic == 0
    d = scene.desc
    assert d.camera_count == 1 and (d.cameras[0].width, d.cameras[0].height) == (8, 6)
    assert d.material_count == 9  # the nine built-ins, src/MaterialLibrary.cpp:6-18
    assert [d.materials[i].name.decode() for i in range(9)] == \
        ["Default", "White", "Black", "Red", "Yellow", "Green", "Cyan", "Blue", "Magenta"]


def test_objects_are_key_sorted_and_parents_compose(rt, tmp_path):
    data = json.loads(json.dumps(BASE))
    data["$materials"] = {"Zed": {"$type": "Mirror", "$color": [0.5]}, "Alpha": {"$type": "Glass", "$ior": 1.33}}
    data["$scene"].update({
        "ZLight": {"$type": "OmniLight", "$brightness": 10, "$color": [1, 1, 1]},
        "ALight": {"$type": "SpotLight", "$brightness": 5, "$angle": 0.45, "$color": [1, 1, 1]},
        "Group": {"$type": "Object", "$position": [10, 0, 0],
                  "$rotation": {"$type": "PitchYawRollDegrees", "$value": [0, 90, 0]},
                  "B": {"$type": "SphereVolume", "$material": "Zed", "$center": [0, 0, 0], "$radius": 2,
                        "$position": [0, 0, 5]},
                  "A": {"$type": "Quad", "$material": "Alpha"}},
        "First": {"$type": "PlaneVolume"},
    })
    scene = rt.Scene(write(tmp_path, data))
    d = scene.desc
    # materials: built-ins then name order; lights and volumes: key order, children after parent
    assert [d.materials[i].name.decode() for i in range(9, 11)] == ["Alpha", "Zed"]
    assert [d.lights[i].name.decode() for i in range(2)] == ["ALight", "ZLight"]
    assert [d.volumes[i].type for i in range(3)] == [3, 2, 1]  # First(plane), Group/A(quad), Group/B(sphere)
    assert d.volumes[1].material == 9 and d.volumes[2].material == 10
    # world transform of Group/B: parent (10,0,0) + R_y(90deg) * (0,0,5) = (15, 0, ~0)
    pos = np.array(list(d.volumes[2].position))
    assert np.allclose(pos, [15, 0, 0], atol=1e-12)
    # spot light flux: I * 2 pi (1 - sin(angle / 2))  (src/SpotLight.cpp:21-44)
    assert d.lights[0].luminous_flux == 5 * ((2 * math.pi) * (1.0 - math.sin(0.45 / 2.0)))
    assert d.lights[0].max_angle == 0.45 / 2.0
    assert d.lights[1].luminous_flux == 10 * (math.pi * 4.0)
    assert d.materials[9].ior == 1.33 and tuple(d.materials[9].color) == (1.0, 1.0, 1.0)  # clear glass default tint


@pytest.mark.parametrize("mutate,message", [
    (lambda d: d["$scene"].update({"X": {"$type": "Teapot"}}), "Unrecognized object type: Teapot"),
    (lambda d: d.update({"$materials": {"M": {"$color": [1]}}}), 'Material "M" requires $type'),
    (lambda d: d.update({"$materials": {"Red": {"$type": "Mirror"}}}), "Library already contains data named Red"),
    (lambda d: d["$scene"].update({"V": {"$type": "SphereVolume", "$material": "Nope"}}),
     "Library does not contain item named Nope"),
    (lambda d: d["$scene"].update({"V": {"$type": "MeshVolume", "$mesh": "Nope"}}),
     "Library does not contain item named Nope"),
    (lambda d: d["$scene"]["Camera"].update({"$position": [1, 2]}),
     "Vector object must be an array with 3 or 4 number elements"),
    (lambda d: d["$scene"]["Camera"].update({"$rotation": {"$type": "Euler", "$value": [0, 0, 0]}}),
     "Unsupported rotation type: Euler"),
    (lambda d: d["$scene"]["Camera"].update({"$projection": "fisheye"}), "Unknown $projection"),
    (lambda d: d["$scene"].update({"Q": {"$type": "Quad", "$corners": [[0, 0, 0]]}}),
     'Quad "$corners" must be an array of exactly 4 vectors'),
    (lambda d: d["$renderConfiguration"].update({"$width": 0}), "Cannot configure Camera with 0 width or 0 height"),
    (lambda d: d["$renderConfiguration"].update({"$photonsPerLight": 0}), "$photonsPerLight must be > 0"),
])
def test_bad_scenes_raise_the_reference_errors(rt, tmp_path, mutate, message):
    data = json.loads(json.dumps(BASE))
    mutate(data)
    with pytest.raises(RuntimeError) as info:
        rt.Scene(write(tmp_path, data))
    assert message in str(info.value)


def test_malformed_and_missing_files_raise(rt, tmp_path):
    with pytest.raises(RuntimeError):
        rt.Scene(write(tmp_path, '{"$scene": { "Camera": '))
    with pytest.raises(RuntimeError):
        rt.Scene(str(tmp_path / "does_not_exist.json"))


def test_stand_in_meshes_and_obj_reader(rt, tmp_path):
    # the reference's mesh files are Git-LFS pointers: named stand-ins are generated
    (tmp_path / "meshes").mkdir()
    (tmp_path / "meshes" / "cube.obj").write_text("version https://git-lfs.github.com/spec/v1\noid sha256:0\nsize 516\n")
    obj = "o Tri\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\n"
    (tmp_path / "meshes" / "quad.obj").write_text(obj)
    data = json.loads(json.dumps(BASE))
    data["$meshes"] = ["meshes/cube.obj", "meshes/quad.obj", "meshes/CornellBox.obj"]
    data["$scene"].update({"C": {"$type": "MeshVolume", "$mesh": "Cube"}, "T": {"$type": "MeshVolume", "$mesh": "Tri"},
                           "W": {"$type": "MeshVolume", "$mesh": "Back"}})
    scene = rt.Scene(write(tmp_path, data))
    d = scene.desc
    names = [d.meshes[i].name.decode() for i in range(d.mesh_count)]
    assert names == ["Cube", "Tri", "Ceiling", "Left", "Right", "Floor", "Back"]
    assert d.meshes[0].triangle_count == 12
    # fan triangulation (v0,v1,v2),(v0,v2,v3)  (tests/test_TriangleWatertight.cpp:15-17)
    t = np.ctypeslib.as_array(d.tri_pos, shape=(d.triangle_count * 9,)).reshape(-1, 3, 3)
    first = int(d.meshes[1].first_triangle)
    assert np.array_equal(t[first], [[0, 0, 0], [1, 0, 0], [1, 1, 0]])
    assert np.array_equal(t[first + 1], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    # an unknown missing mesh file is an error
    data["$meshes"] = ["meshes/absent.obj"]
    with pytest.raises(RuntimeError):
        rt.Scene(write(tmp_path, data, "bad.json"))


def test_config_scenes_load(rt):
    for name in ("CornellBox", "CornellBoxGlassMirrorDiffuse", "MirrorKnot4k", "MirrorBombardment-200M",
                 "MultiCameraTest", "CBGMD_stochastic128", "SinglePhotonVerify", "MirrorDirectTest"):
        scene = rt.Scene(os.path.join(REPO, "scenes", name + ".json"))
        assert scene.desc.volume_count > 0 and scene.desc.light_count > 0 and scene.desc.camera_count > 0
    bomb = rt.Scene(os.path.join(REPO, "scenes", "MirrorBombardment-200M.json"))
    assert bomb.settings.photons_per_light == 200_000_000 and bomb.settings.use_probe_gather == 1
    # object order is alphabetical: MirrorPlate (Cube) before TargetSphere (SphereDiffuse)
    assert [bomb.desc.meshes[bomb.desc.volumes[i].mesh].name.decode() for i in range(2)] == ["Cube", "SphereDiffuse"]
