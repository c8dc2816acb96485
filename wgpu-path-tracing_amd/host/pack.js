'use strict';
/**
 * pack.js — JS objects -> WGSL-layout byte blobs.
 *
 * Stands in for webgpu-utils' makeStructuredView(...).set(...) in the reference
 * (src/renderer/renderer.ts:282-320 for the four storage buffers, :403-413 for the camera):
 * takes the reference's CPU-side types (src/renderer/gpu.ts:10-65, src/renderer/bvh.ts:6-12)
 * and writes the struct layouts of src/shader/pt.wgsl:7-78 (include/ptmi_layout.h).
 * u32 fields go through Uint32Array, so -1 becomes 0xFFFFFFFF and fractional atlas rects
 * truncate, exactly as in the reference.
 */

var TRIANGLE_SIZE = 128, MATERIAL_SIZE = 128, BVH_NODE_SIZE = 48, LIGHT_SIZE = 48, CAMERA_SIZE = 96;

function put3(f32, at, v) { f32[at] = v[0]; f32[at + 1] = v[1]; f32[at + 2] = v[2]; }
function put2(f32, at, v) { f32[at] = v[0]; f32[at + 1] = v[1]; }
function putRect(u32, at, r) {
  r = r || { x: 0, y: 0, w: 0, h: 0 };
  u32[at] = r.x; u32[at + 1] = r.y; u32[at + 2] = r.w; u32[at + 3] = r.h;
}

/** pt.wgsl:28-39 */
function packTriangles(triangles) {
  var buf = new ArrayBuffer(triangles.length * TRIANGLE_SIZE);
  var f = new Float32Array(buf), u = new Uint32Array(buf);
  for (var i = 0; i < triangles.length; i++) {
    var t = triangles[i], b = i * 32;
    put3(f, b + 0, t.v0); put3(f, b + 4, t.v1); put3(f, b + 8, t.v2);
    put3(f, b + 12, t.n0); put3(f, b + 16, t.n1); put3(f, b + 20, t.n2);
    put2(f, b + 24, t.uv0); put2(f, b + 26, t.uv1); put2(f, b + 28, t.uv2);
    u[b + 30] = t.materialIndex;
  }
  return buf;
}

/** pt.wgsl:14-26 */
function packMaterials(materials) {
  var buf = new ArrayBuffer(materials.length * MATERIAL_SIZE);
  var f = new Float32Array(buf), u = new Uint32Array(buf);
  for (var i = 0; i < materials.length; i++) {
    var m = materials[i], b = i * 32;
    put3(f, b + 0, m.baseColor); f[b + 3] = m.metallic; f[b + 4] = m.roughness;
    put3(f, b + 8, m.emission); f[b + 11] = m.emissiveStrength; f[b + 12] = m.ior; f[b + 13] = m.transmission;
    putRect(u, b + 14, m.albedoMap); putRect(u, b + 18, m.normalMap);
    putRect(u, b + 22, m.pbrMap); putRect(u, b + 26, m.emissiveMap);
  }
  return buf;
}

/** pt.wgsl:67-78; node.aabb = {min, max} (src/utils/aabb.ts) */
function packBVH(nodes) {
  var buf = new ArrayBuffer(nodes.length * BVH_NODE_SIZE);
  var f = new Float32Array(buf), u = new Uint32Array(buf);
  for (var i = 0; i < nodes.length; i++) {
    var n = nodes[i], b = i * 12;
    put3(f, b + 0, n.aabb.min); put3(f, b + 4, n.aabb.max);
    u[b + 8] = n.left; u[b + 9] = n.right; u[b + 10] = n.triangleOffset; u[b + 11] = n.triangleCount;
  }
  return buf;
}

/** pt.wgsl:45-51 */
function packLights(lights) {
  var buf = new ArrayBuffer(lights.length * LIGHT_SIZE);
  var f = new Float32Array(buf), u = new Uint32Array(buf);
  for (var i = 0; i < lights.length; i++) {
    var l = lights[i], b = i * 12;
    put3(f, b + 0, l.position); u[b + 3] = l.lightType; put3(f, b + 4, l.color); f[b + 7] = l.intensity;
    u[b + 8] = l.triangleIndex;
  }
  return buf;
}

/** pt.wgsl:53-65 — the 96-byte uniform rewritten every frame (renderer.ts:403-413) */
function packCamera(camera, out) {
  var buf = out || new ArrayBuffer(CAMERA_SIZE);
  var f = new Float32Array(buf), u = new Uint32Array(buf);
  put3(f, 0, camera.position); put3(f, 4, camera.forward); put3(f, 8, camera.right); put3(f, 12, camera.up);
  f[15] = camera.fov; f[16] = camera.aspect; u[17] = camera.width; u[18] = camera.height; u[19] = camera.frameIndex;
  f[20] = camera.aperture; f[21] = camera.focusDistance;
  return buf;
}

/** SceneData (gpu.ts:60-65) -> the four blobs of bindings 1, 2, 4, 5 */
function packScene(sceneData) {
  return {
    triangles: packTriangles(sceneData.triangles), materials: packMaterials(sceneData.materials),
    bvhNodes: packBVH(sceneData.bvhNodes), lights: packLights(sceneData.lights),
  };
}

module.exports = {
  TRIANGLE_SIZE: TRIANGLE_SIZE, MATERIAL_SIZE: MATERIAL_SIZE, BVH_NODE_SIZE: BVH_NODE_SIZE, LIGHT_SIZE: LIGHT_SIZE,
  CAMERA_SIZE: CAMERA_SIZE, packTriangles: packTriangles, packMaterials: packMaterials, packBVH: packBVH,
  packLights: packLights, packCamera: packCamera, packScene: packScene,
};
