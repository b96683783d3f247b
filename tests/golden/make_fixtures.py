#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.  Run in the build container (the only
place where /root/reference and oracle/_ref exist):  python tests/golden/make_fixtures.py

  *_tinyobj.json          output of the REAL tinyobjloader (oracle/_ref/tinyobj_dump, compiled from the
                          reference tree) on this repo's OBJ files -> pins csrc/host/obj_loader.hpp
  obj_hashes.json         sha256 of the same dump for the big procedural scenes
  cornell_ref_ingest.json what the reference's ingest semantics (loadOBJandMTL on the reference's own
                          external/tinyobjloader/models/cornell_box.obj) yields through the real loader:
                          per-shape counts/offsets and a position checksum
  cornell_256_oracle.npz  oracle (brute force, no BVH) render of Cornell 256x256 1spp: the 5 ray-gen images
                          + HDR — a regression pin of the oracle itself (parity is otherwise UNPINNED: the
                          reference ships no golden image)
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
DUMP = os.path.join(ROOT, "oracle", "_ref", "tinyobj_dump")
REF_CORNELL = "/root/reference/external/tinyobjloader/models/cornell_box.obj"


def dump(obj, mtl=""):
    return subprocess.check_output([DUMP, obj, mtl]).decode()


def ingest_semantics(j):
    """loadOBJandMTL (reference src/core/file.cppm:130-268) applied to a tinyobj dump: per-shape de-dup."""
    V, N, T = j["vertices"], j["normals"], j["texcoords"]
    shapes, voff, ioff = [], 0, 0
    pos_sum = 0.0
    for s in j["shapes"]:
        uniq, idx = {}, []
        off = 0
        for fv in s["num_face_vertices"]:
            if fv == 3:
                for k in range(3):
                    vi, ni, ti = s["indices"][off + k]
                    key = (tuple(V[3 * vi:3 * vi + 3]), tuple(N[3 * ni:3 * ni + 3]) if ni >= 0 else (0.0, 0.0, 0.0),
                           tuple(T[2 * ti:2 * ti + 2]) if ti >= 0 else (0.0, 0.0))
                    if key not in uniq:
                        uniq[key] = len(uniq)
                        pos_sum += sum(key[0])
                    idx.append(uniq[key])
            off += fv
        shapes.append({"name": s["name"], "vertices": len(uniq), "indices": len(idx), "vertexOffset": voff, "indexOffset": ioff,
                       "material_id0": s["material_ids"][0] if s["material_ids"] else -1, "index_list": idx})
        voff += len(uniq)
        ioff += len(idx)
    return {"shapes": shapes, "totalVertices": voff, "totalIndices": ioff, "positionSum": pos_sum,
            "materials": [{"name": m["name"], "diffuse": m["diffuse"], "specular": m["specular"]} for m in j["materials"]]}


def image_hashes():
    """4) image ingest: every file of tests/image_cases.py through the REAL stb_image of the reference tree
    (oracle/_ref/stb_dump: flip on load, STBI_rgb_alpha and STBI_grey) -> tests/golden/image_stb_hashes.json"""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from image_cases import image_cases
    stb = os.path.join(ROOT, "oracle", "_ref", "stb_dump")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, data in image_cases():
            p = os.path.join(d, name)
            open(p, "wb").write(data)
            out[name] = {"file_sha256": hashlib.sha256(data).hexdigest()}
            for want in (4, 1):
                raw = subprocess.run([stb, p, str(want)], check=True, capture_output=True).stdout
                head, _, body = raw.partition(b"\n")
                w, h, c = (int(x) for x in head.split())
                assert len(body) == w * h * c
                e = {"shape": [h, w, c], "sha256": hashlib.sha256(body).hexdigest()}
                if name.endswith(".hdr"):
                    e["bytes_hex"] = body.hex()          # compared with a tolerance of one code value (pow() in the hdr -> ldr step)
                out[name][str(want)] = e
    json.dump(out, open(os.path.join(HERE, "image_stb_hashes.json"), "w"), indent=1)
    print("image_stb_hashes.json:", len(out), "files")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "images":
        return image_hashes()
    from realtimeraytracer_amd import _abi as A
    from realtimeraytracer_amd import scenes
    from oracle import oracle_py as O

    os.environ["RTR_SCENE_CACHE"] = "/tmp/rtr_fixture_scenes"
    os.makedirs("/tmp/rtr_fixture_scenes", exist_ok=True)

    # 1) real-loader dumps of this repo's OBJ files
    cwd = os.getcwd()
    os.chdir(HERE)
    open("features_tinyobj.json", "w").write(dump("features.obj"))
    os.chdir(cwd)
    cobj, cdir = scenes.write_cornell("/tmp/rtr_fixture_scenes")
    open(os.path.join(HERE, "cornell_tinyobj.json"), "w").write(dump(cobj, cdir))
    hashes = {}
    for name, writer in (("bunny_class", scenes.write_bunny_class), ("sponza_class", scenes.write_sponza_class)):
        obj, d = writer("/tmp/rtr_fixture_scenes")
        text = dump(obj, d)
        hashes[name] = {"tinyobj_dump_sha256": hashlib.sha256(text.encode()).hexdigest(), "obj_sha256": scenes.file_sha256(obj),
                        "triangles": sum(len(s["num_face_vertices"]) for s in json.loads(text)["shapes"])}
    json.dump(hashes, open(os.path.join(HERE, "obj_hashes.json"), "w"), indent=1)

    # 2) the reference's own Cornell model through the real loader + the reference's ingest semantics
    if os.path.exists(REF_CORNELL):
        j = json.loads(dump(REF_CORNELL, os.path.dirname(REF_CORNELL) + "/"))
        json.dump(ingest_semantics(j), open(os.path.join(HERE, "cornell_ref_ingest.json"), "w"), indent=1)

    # 3) oracle pin: Cornell 256x256, 1 spp, all five ray-gen images (synthetic LTC tables) + HDR, brute force
    s = scenes.cornell_box(256, 256, ltc=scenes.synthetic_ltc())
    p = A.rtr_render_params(256, 256, 1, 3, A.IMAGES_RAYGEN5 | A.IMG_BIT(A.IMAGE_HDR), 8, 0, 1, 0, 0, 1, 0)
    r = O.render(s.desc, s.camera, s.scene_info(0), p, bvh=None, images=A.IMAGES_RAYGEN5 | A.IMG_BIT(A.IMAGE_HDR), threads=8)
    np.savez_compressed(os.path.join(HERE, "cornell_256_oracle.npz"), analytic=r.images[0], shadowed=r.images[1], unshadowed=r.images[2],
                        normal=r.images[6], position=r.images[7], hdr=r.hdr,
                        counters=np.array([r.stats.numRays, r.stats.numPrimaryRays, r.stats.numShadowRays, r.stats.numHits,
                                           r.stats.numLightFetches, r.stats.numLightTriFetches], dtype=np.uint64))
    image_hashes()
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
