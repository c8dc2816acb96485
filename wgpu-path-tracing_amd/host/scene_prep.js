'use strict';
/**
 * scene_prep.js — glTF -> SceneData, following src/renderer/gpu.ts:67-421 step by step:
 * world matrices by walking up the parent chain (:77-103), per node normal matrix, punctual lights
 * and mesh primitives in node order (:194-299), one material per primitive (:285-291, :356-421),
 * then the SAH-BVH (which sorts the triangles, bvh.ts) and one emissive light per emissive triangle
 * in post-sort order (:121-138). The BVH build and the light list run in libptmi_scene.so through
 * the addon; the result is the four WGSL-layout blobs plus the SceneData-style counts.
 */
var path = require('path');
var M = require('./mat');
var pack = require('./pack');

function addon() { return require(path.join(__dirname, 'addon', 'ptmi_napi.node')); }

/** gpu.ts:153-191 */
function extractNodeMatrix(node) {
  var matrix = node.matrix ? M.create16(node.matrix) : M.identity();
  if (!node.matrix) {
    if (node.translation) M.translate(matrix, M.vec3(node.translation[0], node.translation[1], node.translation[2]), matrix);
    if (node.rotation) {
      var q = new Float32Array(4);
      q[0] = node.rotation[0]; q[1] = node.rotation[1]; q[2] = node.rotation[2]; q[3] = node.rotation[3];
      M.mul(matrix, M.fromQuat(q), matrix);
    }
    if (node.scale) M.scale(matrix, M.vec3(node.scale[0], node.scale[1], node.scale[2]), matrix);
  }
  return matrix;
}

var ZERO_RECT = { x: 0, y: 0, w: 0, h: 0 };

/** gpu.ts:356-421; atlas = Map(material -> {albedoMap, normalMap, pbrMap, emissiveMap}) from atlas.js packing() */
function buildMaterial(material, atlas) {
  if (!material) {
    return {
      baseColor: [1, 1, 1], emission: [0, 0, 0], emissiveStrength: 0.0, metallic: 0.0, roughness: 0.1, ior: 1.5,
      transmission: 0.0, albedoMap: ZERO_RECT, normalMap: ZERO_RECT, pbrMap: ZERO_RECT, emissiveMap: ZERO_RECT,
    };
  }
  var pbr = material.pbrMetallicRoughness || {};
  var ext = material.extensions || {};
  var def = function (v, d) { return v === undefined || v === null ? d : v; };
  var baseColor = def(pbr.baseColorFactor, [1, 1, 1, 1]);
  var emissive = def(material.emissiveFactor, [0, 0, 0]);
  var rects = (atlas && atlas.get(material)) || {};
  return {
    baseColor: [baseColor[0], baseColor[1], baseColor[2]],
    metallic: def(pbr.metallicFactor, 1.0), roughness: def(pbr.roughnessFactor, 1.0),
    emission: [emissive[0], emissive[1], emissive[2]],
    emissiveStrength: def((ext.KHR_materials_emissive_strength || {}).emissiveStrength, 1.0),
    ior: def((ext.KHR_materials_ior || {}).ior, 1.5),
    transmission: def((ext.KHR_materials_transmission || {}).transmissionFactor, 0.0),
    albedoMap: rects.albedoMap || ZERO_RECT, normalMap: rects.normalMap || ZERO_RECT,
    pbrMap: rects.pbrMap || ZERO_RECT, emissiveMap: rects.emissiveMap || ZERO_RECT,
  };
}

/** gpu.ts:301-354 (indexed meshes; a non-indexed mesh throws there too) */
function buildTriangles(position, normal, uv, index) {
  if (!index) throw new Error('No index found');
  uv = uv || new Float32Array(position.length);
  var tris = [];
  for (var i = 0; i < index.length; i += 3) {
    var i0 = index[i] * 3, i1 = index[i + 1] * 3, i2 = index[i + 2] * 3;
    var u0 = index[i] * 2, u1 = index[i + 1] * 2, u2 = index[i + 2] * 2;
    tris.push({
      v0: [position[i0], position[i0 + 1], position[i0 + 2]], v1: [position[i1], position[i1 + 1], position[i1 + 2]],
      v2: [position[i2], position[i2 + 1], position[i2 + 2]],
      n0: [normal[i0], normal[i0 + 1], normal[i0 + 2]], n1: [normal[i1], normal[i1 + 1], normal[i1 + 2]],
      n2: [normal[i2], normal[i2 + 1], normal[i2 + 2]],
      uv0: [uv[u0], uv[u0 + 1]], uv1: [uv[u1], uv[u1 + 1]], uv2: [uv[u2], uv[u2 + 1]], materialIndex: 0,
    });
  }
  return tris;
}

/** gpu.ts:194-299 */
function processNode(gltf, node, allTriangles, allMaterials, allLights, world, atlas) {
  var normalMat = M.transpose(M.inverse(world));
  if (node.light !== undefined) {
    var light = gltf.lights[node.light];
    var color = light.color ? [light.color[0], light.color[1], light.color[2]] : [1, 1, 1];
    var intensity = light.intensity === undefined || light.intensity === null ? 1.0 : light.intensity;
    if (light.type === 'directional') {
      var rot = M.quatFromMat(world);
      allLights.push({ position: M.transformQuat(M.vec3(0, 0, -1), rot), lightType: 1, color: color, intensity: intensity, triangleIndex: 0 });
    } else if (light.type === 'point') {
      allLights.push({ position: M.transformMat4(M.vec3(0, 0, 0), world), lightType: 2, color: color, intensity: intensity, triangleIndex: 0 });
    }
  }
  if (node.mesh) {
    node.mesh.primitives.forEach(function (prim) {
      var position = prim.attributes.POSITION.value, normal = prim.attributes.NORMAL.value;
      var uv = prim.attributes.TEXCOORD_0 ? prim.attributes.TEXCOORD_0.value : undefined;
      var tp = new Float32Array(position.length), tn = new Float32Array(normal.length);
      for (var i = 0; i < position.length; i += 3) {
        var p = M.transformMat4(M.vec3(position[i], position[i + 1], position[i + 2]), world);
        tp[i] = p[0]; tp[i + 1] = p[1]; tp[i + 2] = p[2];
        var n = M.normalize(M.transformMat4Upper3x3(M.vec3(normal[i], normal[i + 1], normal[i + 2]), normalMat));
        tn[i] = n[0]; tn[i + 1] = n[1]; tn[i + 2] = n[2];
      }
      var tris = buildTriangles(tp, tn, uv, prim.indices ? prim.indices.value : undefined);
      allMaterials.push(buildMaterial(prim.material, atlas));
      tris.forEach(function (t) { t.materialIndex = allMaterials.length - 1; allTriangles.push(t); });
    });
  }
}

/** loader.ts:20-40 + gpu.ts:67-150 -> { blobs, atlas, counts, bvhDepth } */
function prepareScene(gltf) {
  var packed = require('./atlas').packing(gltf);
  var allTriangles = [], allMaterials = [], allLights = [];
  var parent = new Map();
  gltf.nodes.forEach(function (n) { (n.children || []).forEach(function (c) { parent.set(c, n); }); });
  var world = new Map();
  gltf.nodes.forEach(function (node) {
    var w = M.clone(extractNodeMatrix(node));
    var cur = node;
    while (parent.has(cur)) { cur = parent.get(cur); M.mul(extractNodeMatrix(cur), w, w); }
    world.set(node, w);
  });
  gltf.nodes.forEach(function (node) { processNode(gltf, node, allTriangles, allMaterials, allLights, world.get(node), packed.materials); });

  var triangles = pack.packTriangles(allTriangles);             // sorted in place by the BVH build
  var materials = pack.packMaterials(allMaterials);
  var a = addon();
  var bvh = a.buildBvh(triangles);
  var lights = a.emissiveLights(triangles, materials, pack.packLights(allLights));
  return {
    blobs: { triangles: triangles, materials: materials, bvhNodes: bvh.nodes, lights: lights },
    atlas: packed.texture,
    counts: { triangles: allTriangles.length, materials: allMaterials.length, bvhNodes: bvh.nodes.byteLength / pack.BVH_NODE_SIZE,
      lights: lights.byteLength / pack.LIGHT_SIZE, punctualLights: allLights.length },
    bvhDepth: bvh.depth,
  };
}

module.exports = { prepareScene: prepareScene, extractNodeMatrix: extractNodeMatrix, buildMaterial: buildMaterial };
