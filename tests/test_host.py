"""Host layer: the reference-API mirror describes scenes into the flat RtSceneDesc the
way the reference builds them (node order, transforms, constructor-time normalisations,
tobj-style OBJ re-indexing)."""
import ctypes as C
import math

import numpy as np
import pytest

RT_NONE = 0xFFFFFFFF


def desc_of(scene, abi_mod):
    return C.cast(scene.desc, C.POINTER(abi_mod.RtSceneDesc)).contents


def test_glass_monkey_node_order_and_materials(scene_cache, abi_mod):
    d = desc_of(scene_cache("glass_monkey"), abi_mod)
    # DFS pre-order: 0 root, 1 room, 2 ceiling, 3 floor, 4 front, 5 back, 6 left, 7 right, 8 monkey
    assert d.num_nodes == 9
    assert [d.nodes[i].parent for i in range(9)] == [RT_NONE, 0, 1, 1, 1, 1, 1, 1, 0]
    assert [d.nodes[i].subtree_size for i in range(9)] == [9, 7, 1, 1, 1, 1, 1, 1, 1]
    assert d.nodes[0].prim == RT_NONE and d.nodes[1].prim == RT_NONE
    for i in range(2, 8):
        p = d.prims[d.nodes[i].prim]
        assert p.kind == 1 and p.p[0] == pytest.approx(707.0) and p.p[1] == pytest.approx(707.0)   # Rectangle(700*1.01)
    assert d.prims[d.nodes[8].prim].kind == 10   # mesh
    # ceiling: rotation(X, 90) * translation(0,0,-350) puts the wall at y = +350
    m = np.array(list(d.nodes[2].trans.m)).reshape(4, 4).T   # column-major → rows
    assert m[:3, 3] == pytest.approx([0.0, 350.0, 0.0], abs=1e-9)
    # glass(): Composite[(0.1, Reflection(WHITE)), (0.9, Translucent(WHITE, 1.517))]
    g = d.shaders[d.nodes[8].shader]
    assert g.kind == 5 and g.num_children == 2
    kids = [d.shader_child_idx[g.first_child + k] for k in range(2)]
    assert [d.shader_child_weight[g.first_child + k] for k in range(2)] == [0.1, 0.9]
    assert d.shaders[kids[0]].kind == 2 and d.shaders[kids[1]].kind == 3
    assert d.shaders[kids[1]].p[3] == 1.517
    # three point lights, power 150000, falloff (0, 0, pi)
    assert d.num_lights == 3
    assert all(d.lights[i].kind == 0 and d.lights[i].power == 150000.0 for i in range(3))
    assert d.lights[0].falloff[2] == pytest.approx(math.pi)


def test_obj_reader_reindexes_like_tobj(scene_cache, abi_mod):
    d = desc_of(scene_cache("glass_monkey"), abi_mod)
    m = d.meshes[0]
    # monkey2.obj: 507 v / 523 vt / 507 vn, 967 triangular faces → 523 distinct (v,vt,vn) corners
    assert (m.num_faces, m.num_positions, m.num_normals, m.num_texcoords) == (967, 523, 523, 523)
    faces = np.array([d.mesh_u32[m.faces_off + i] for i in range(3 * m.num_faces)])
    assert faces.max() == m.num_positions - 1
    # indices are handed out in order of first appearance
    first_seen = []
    for f in faces:
        if f not in first_seen[-600:] and f == len(first_seen):
            first_seen.append(f)
    assert len(first_seen) == m.num_positions
    # positions are f32 values widened to f64
    pos = np.array([d.mesh_f64[m.positions_off + i] for i in range(3 * m.num_positions)])
    assert np.array_equal(pos, pos.astype(np.float32).astype(np.float64))


def test_light_constructor_normalises_colour(scene_cache, abi_mod):
    d = desc_of(scene_cache("basic_setup_scene"), abi_mod)
    # PointLight::new(.., Color(0,1,1), ..): max channel → 1 (already), kept as is
    assert list(d.lights[0].color) == [0.0, 1.0, 1.0]
    d2 = desc_of(scene_cache("light_room"), abi_mod)
    assert d2.lights[0].kind == 1 and d2.lights[0].size == 500.0 and d2.lights[0].power == 1500000.0


def test_light_room_layout(scene_cache, abi_mod):
    d = desc_of(scene_cache("light_room"), abi_mod)
    assert d.num_nodes == 10
    assert d.prims[d.nodes[8].prim].kind == 0 and d.prims[d.nodes[8].prim].p[0] == 80.0     # sphere r=80
    assert d.prims[d.nodes[9].prim].kind == 5 and d.prims[d.nodes[9].prim].p[0] == 160.0    # cube 160
    mix = d.shaders[d.nodes[9].shader]
    assert mix.kind == 4 and mix.num_children == 2                                          # Mix[Texture, Phong]
    tex = d.shaders[d.shader_child_idx[mix.first_child]]
    assert tex.kind == 1
    im = d.images[tex.ref]
    assert (im.width, im.height, im.channels) == (1024, 768, 3)
    assert d.shaders[d.nodes[7].shader].kind == 2                                           # right wall: mirror


def test_csg_description(scene_cache, abi_mod):
    d = desc_of(scene_cache("subtraction_outside1"), abi_mod)
    assert d.has_background == 1
    kinds = [d.csg[i].kind for i in range(d.num_csg)]
    # spikey thing: 4 base spheres under 3 nested subtractions; weird cube: cube - sphere
    assert kinds.count(0) == 6 and kinds.count(1) == 4
    root_prims = [d.prims[d.nodes[i].prim] for i in (1, 2)]
    assert all(p.kind == 11 for p in root_prims)


def test_unknown_scene_is_an_error(scenes_mod):
    with pytest.raises(RuntimeError):
        scenes_mod.Scene("no_such_scene")
