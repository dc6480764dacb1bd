#!/usr/bin/env python3
"""Offline asset preparation (run once in the authoring container, outputs committed).

Decodes the reference's image assets with PIL into headered raw files the C++ host
layer can read without an image library, and copies the OBJ model and the gallery
golden.  Raw format (.rtim): 4-byte magic b"RTIM", u32 width, u32 height, u32 channels
(little endian), then width*height*channels bytes, rows top-first.

JPEG sources: PIL/libjpeg's IDCT may differ from the reference's jpeg-decoder 0.1.15
by +-1 per texel; PNG sources are exact.  Both the oracle and the device path read
these same raw files.
"""
import os, shutil, struct, sys
from PIL import Image

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")

IMAGES = [
    "textures/cube_rgb_gradient.png",   # light_room cube (tests/square_light_test.rs:88)
    "textures/test3.png",               # mesh_basic (tests/mesh_test.rs:73)
    "textures/marble.png",              # ceramic_tile (tests/mesh_test.rs:142)
    "textures/smooth_concrete.jpg",     # concrete (tests/mesh_test.rs:155)
    "bump_maps/bumps.png",
    "bump_maps/dot.png",
    "bump_maps/tiles.jpg",              # ceramic_tile bump map (tests/mesh_test.rs:138)
    "normal_maps/concrete.jpg",         # concrete / metallic / frosted_glass
    "backgrounds/sky_ocean.jpg",        # substitute for the missing forest2.jpg / building.jpg
    "backgrounds/room_lores.jpg",
    "tiny.png",
    "textures/d8_num.png",              # octahedron_basic (tests/polyhedra_test.rs:90)
    "textures/d6_num.png",              # dice_scene d6 (tests/dice_scene.rs:56)
    "textures/d4_num.png",              # dice_scene d4, tetrahedron_basic (tests/dice_scene.rs:66)
    "bump_maps/d6_num.png",             # dice_scene d6 bump map (tests/dice_scene.rs:53)
    "textures/cube_rgb_numbers.png",    # texture_cube (tests/texture_test.rs:88)
    "textures/keanu.jpg",               # plane_texture (tests/texture_test.rs:154); stands in for the 11 MB gurf-1.jpg
    "textures/light_wood.jpg",          # dice_scene floor (tests/dice_scene.rs:33); stands in for the 35 MB wood_boards.jpg
    "living_room.jpg",                  # skybox_test (tests/texture_test.rs:104)
    "normal_maps/brick_wall_01.png",    # stands in for the missing normal_maps/brick_wall.jpg (src/util.rs:17)
]

def convert(rel):
    src = os.path.join(REF, "assets/images", rel)
    dst = os.path.join(OUT, "images", os.path.splitext(rel)[0] + ".rtim")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    im = Image.open(src).convert("RGB")
    w, h = im.size
    with open(dst, "wb") as f:
        f.write(b"RTIM" + struct.pack("<III", w, h, 3))
        f.write(im.tobytes())
    print(f"{rel}: {w}x{h} -> {dst}")

for rel in IMAGES:
    convert(rel)
os.makedirs(os.path.join(OUT, "models"), exist_ok=True)
for m in ["monkey2.obj", "cube.obj"]:
    shutil.copyfile(os.path.join(REF, "assets/models", m), os.path.join(OUT, "models", m))
gold = os.path.join(OUT, "..", "tests", "golden")
os.makedirs(gold, exist_ok=True)
for g in ["glass_monkey", "light_room", "monkey_1"]:
    shutil.copyfile(os.path.join(REF, f"gallery/{g}.png"), os.path.join(gold, f"gallery_{g}.png"))
