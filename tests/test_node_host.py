"""The Node host (wgpu-path-tracing_amd/host): byte-exact packing of the reference's CPU-side types,
addon loading, and (GPU) the whole JS -> N-API -> C ABI -> HIP chain against the oracle."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from ptmi import layout, scene_io, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "wgpu-path-tracing_amd", "host")
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


def _build_addon():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "wgpu-path-tracing_amd"), "all"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(HOST, "addon")], stdout=subprocess.DEVNULL)


def test_js_packing_is_byte_exact(tmp_path):
    """pack.js (the webgpu-utils stand-in, renderer.ts:282-320, :403-413) == the numpy layouts."""
    sc = scenes.make("feature_box")            # has atlas rects, -1 children, all three light types
    cam = layout.make_camera(640, 360, frame_index=7, aperture=0.05, focus_distance=2.8)
    (tmp_path / "scene.json").write_text(json.dumps(scene_io.scene_data_json(sc, cam)))
    subprocess.check_call([NODE, os.path.join(HOST, "pack_cli.js"), str(tmp_path / "scene.json"), str(tmp_path)])
    for name, arr in (("triangles", sc.tris), ("materials", sc.mats), ("bvhNodes", sc.nodes), ("lights", sc.lights)):
        assert (tmp_path / f"{name}.bin").read_bytes() == arr.tobytes(), name
    assert (tmp_path / "camera.bin").read_bytes() == cam.tobytes()


def test_ptscene_roundtrip_in_node(tmp_path):
    sc = scenes.make("feature_box")
    scene_io.save_ptscene(sc, str(tmp_path / "s.ptscene"))
    js = ("var f=require(%r).readSceneFile(%r);var c=require('crypto');"
          "var h=function(b){return c.createHash('sha1').update(Buffer.from(b)).digest('hex')};"
          "console.log(JSON.stringify({t:h(f.blobs.triangles),m:h(f.blobs.materials),n:h(f.blobs.bvhNodes),"
          "l:h(f.blobs.lights),a:h(f.atlas.data),w:f.atlas.width,fmt:f.atlas.format}))"
          % (os.path.join(HOST, "scene_file.js"), str(tmp_path / "s.ptscene")))
    out = json.loads(subprocess.check_output([NODE, "-e", js]))
    import hashlib
    h = lambda a: hashlib.sha1(a.tobytes()).hexdigest()
    assert out == {"t": h(sc.tris), "m": h(sc.mats), "n": h(sc.nodes), "l": h(sc.lights), "a": h(sc.atlas),
                   "w": 256, "fmt": 1}


def test_addon_loads_and_fails_loudly_without_gpu():
    _build_addon()
    js = ("var h=require(%r);try{var r=new h.Renderer({width:8,height:8});console.log('CREATED');r.destroy();}"
          "catch(e){console.log('ERR '+e.message)}" % os.path.join(HOST, "renderer.js"))
    out = subprocess.check_output([NODE, "-e", js], text=True).strip()
    assert out == "CREATED" or ("ptmi_create failed" in out and "no CPU backend" in out), out


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 4])
def test_node_render_matches_oracle(tmp_path, oracle, batch):
    """BASELINE configs[0] shape (Cornell, 16 spp, 4 bounces, MIS off) through the JS host, reduced to 96x96."""
    _build_addon()
    sc = scenes.make("cornell")
    scene_io.save_ptscene(sc, str(tmp_path / "cornell.ptscene"))
    W = H = 96
    out = subprocess.check_output([NODE, os.path.join(HOST, "render_cli.js"), str(tmp_path / "cornell.ptscene"),
                                   str(tmp_path / "out.f32"), "--width", str(W), "--height", str(H), "--frames", "16",
                                   "--bounces", "4", "--mis", "0", "--batch", str(batch)], text=True)
    st = json.loads(out.strip().splitlines()[-1])
    got = np.fromfile(tmp_path / "out.f32", np.float32).reshape(H, W, 4)
    ref, ost = oracle.render(sc, layout.make_camera(W, H), 16, max_bounces=4, do_mis=0)
    assert st["segments"] == ost.segments and st["paths"] == ost.paths
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_node_blit_png(tmp_path, oracle):
    """JS host -> addon.blit -> PNG: the tone-mapped canvas matches the oracle's blit of the same buffer."""
    from PIL import Image
    _build_addon()
    sc = scenes.make("cornell")
    scene_io.save_ptscene(sc, str(tmp_path / "cornell.ptscene"))
    W, H = 128, 72
    subprocess.check_output([NODE, os.path.join(HOST, "render_cli.js"), str(tmp_path / "cornell.ptscene"),
                             str(tmp_path / "out.f32"), "--width", str(W), "--height", str(H), "--frames", "8",
                             "--batch", "8", "--png", str(tmp_path / "out.png")], text=True)
    hdr = np.fromfile(tmp_path / "out.f32", np.float32).reshape(H, W, 4)
    png = np.array(Image.open(tmp_path / "out.png"))
    ref8 = (np.clip(np.nan_to_num(oracle.blit(hdr), nan=0.0), 0, 1) * 255 + 0.5).astype(np.uint8)
    assert png.shape == (H, W, 4) and (png[..., 3] == 255).all()
    assert (png[..., :3] == ref8[..., :3]).all(axis=-1).mean() >= 0.999


@pytest.mark.gpu
def test_node_blit_rejects_a_buffer_of_the_wrong_size(tmp_path):
    """addon.blit writes width*height*4 bytes: a short array, or one kept from before resize(), must raise a RangeError
    in JS instead of reaching the library (which would write past the end of the V8 buffer)."""
    _build_addon()
    sc = scenes.make("cornell")
    scene_io.save_ptscene(sc, str(tmp_path / "c.ptscene"))
    js = ("var h=require(%r);var r=new h.Renderer({width:16,height:8});r.loadModel(%r).then(function(){"
          "r.renderFrame();var res=[];var ok=r.blit();res.push(ok.length);"
          "[new Uint8Array(10),new Uint8Array(16*8*4+4)].forEach(function(b){try{r.addon.blit(r.ctx,b);res.push('no error')}"
          "catch(e){res.push(e instanceof RangeError?'RangeError':String(e))}});"
          "r.resize(32,8);try{r.addon.blit(r.ctx,ok);res.push('no error')}catch(e){res.push(e instanceof RangeError?'RangeError':String(e))}"
          "res.push(r.blit().length);console.log(JSON.stringify(res));r.destroy();})"
          % (os.path.join(HOST, "renderer.js"), str(tmp_path / "c.ptscene")))
    out = json.loads(subprocess.check_output([NODE, "-e", js], text=True).strip().splitlines()[-1])
    assert out == [16 * 8 * 4, "RangeError", "RangeError", "RangeError", 32 * 8 * 4]
