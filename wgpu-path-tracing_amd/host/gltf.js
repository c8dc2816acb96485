'use strict';
/**
 * gltf.js — minimal binary-glTF (.glb) reader for the host: what the reference gets from
 * @loaders.gl/gltf's load + postProcessGLTF (src/renderer/loader.ts:13-17), reduced to the fields
 * src/renderer/gpu.ts reads: nodes (children, matrix | TRS, mesh, light), mesh primitives with
 * POSITION / NORMAL / TEXCOORD_0 / indices as typed arrays, materials, KHR_lights_punctual lights.
 * Images are decoded to RGBA8 ({width, height, data}; PNG and JPEG) and texture
 * references are resolved the way postProcessGLTF does (`material.normalTexture.texture.source.image`), which is
 * what src/renderer/atlas.ts:36-48 walks.
 */
var fs = require('fs');
var decodePNG = require('./png_decode').decodePNG;
var decodeJPEG = require('./jpeg_decode').decodeJPEG;

var COMPONENTS = { 5120: Int8Array, 5121: Uint8Array, 5122: Int16Array, 5123: Uint16Array, 5125: Uint32Array, 5126: Float32Array };
var COUNTS = { SCALAR: 1, VEC2: 2, VEC3: 3, VEC4: 4, MAT2: 4, MAT3: 9, MAT4: 16 };

function parseGLB(buf) {
  if (buf.readUInt32LE(0) !== 0x46546c67) throw new Error('not a GLB file');
  if (buf.readUInt32LE(4) !== 2) throw new Error('unsupported glTF version');
  var total = buf.readUInt32LE(8), off = 12, json = null, bin = null;
  while (off + 8 <= total) {
    var len = buf.readUInt32LE(off), type = buf.readUInt32LE(off + 4);
    var data = buf.slice(off + 8, off + 8 + len);
    if (type === 0x4e4f534a) json = JSON.parse(data.toString('utf8'));
    else if (type === 0x004e4942 && !bin) bin = data;
    off += 8 + len;
  }
  if (!json) throw new Error('GLB without a JSON chunk');
  return { json: json, bin: bin };
}

/** accessor -> { value: TypedArray (tightly packed copy), size: components per element } */
function readAccessor(json, bin, index) {
  var acc = json.accessors[index], view = json.bufferViews[acc.bufferView];
  var T = COMPONENTS[acc.componentType], n = COUNTS[acc.type];
  if (!T || !n) throw new Error('unsupported accessor type');
  if (view.buffer !== 0 || !bin) throw new Error('only the GLB-embedded buffer is supported');
  var base = bin.byteOffset + (view.byteOffset || 0) + (acc.byteOffset || 0);
  var esz = T.BYTES_PER_ELEMENT, stride = view.byteStride || n * esz;
  var out = new T(acc.count * n);
  var dv = new DataView(bin.buffer, base);
  var get = { 5120: 'getInt8', 5121: 'getUint8', 5122: 'getInt16', 5123: 'getUint16', 5125: 'getUint32', 5126: 'getFloat32' }[acc.componentType];
  for (var i = 0; i < acc.count; i++) for (var k = 0; k < n; k++) out[i * n + k] = dv[get](i * stride + k * esz, true);
  return { value: out, size: n };
}

/** bytes of image i: a GLB bufferView or a base64 data URI */
function imageBytes(json, bin, img) {
  if (img.bufferView !== undefined) {
    var view = json.bufferViews[img.bufferView];
    if (view.buffer !== 0 || !bin) throw new Error('only the GLB-embedded buffer is supported');
    return bin.slice(view.byteOffset || 0, (view.byteOffset || 0) + view.byteLength);
  }
  var m = /^data:[^;,]*;base64,(.*)$/.exec(img.uri || '');
  if (!m) throw new Error('image "' + (img.name || '') + '": external files are not supported');
  return Buffer.from(m[1], 'base64');
}

function decodeImage(bytes, name) {
  if (bytes.length >= 8 && bytes[0] === 137 && bytes[1] === 80) return decodePNG(bytes);
  if (bytes.length >= 2 && bytes[0] === 0xff && bytes[1] === 0xd8) return decodeJPEG(bytes);
  throw new Error('image "' + name + '": unknown format');
}

/** The post-processed shape gpu.ts and atlas.ts consume: index references resolved to objects. */
function loadGLB(pathOrBuffer) {
  var buf = typeof pathOrBuffer === 'string' ? fs.readFileSync(pathOrBuffer) : pathOrBuffer;
  var glb = parseGLB(buf), json = glb.json, bin = glb.bin;
  var images = (json.images || []).map(function (img, i) {
    return { name: img.name, mimeType: img.mimeType, image: decodeImage(imageBytes(json, bin, img), img.name || String(i)) };
  });
  var textures = (json.textures || []).map(function (t) {
    return { name: t.name, sampler: t.sampler, source: t.source !== undefined ? images[t.source] : undefined };
  });
  var resolve = function (info) { if (info && info.index !== undefined) info.texture = textures[info.index]; };
  var materials = (json.materials || []).map(function (m) {
    resolve(m.normalTexture); resolve(m.occlusionTexture); resolve(m.emissiveTexture);
    if (m.pbrMetallicRoughness) { resolve(m.pbrMetallicRoughness.baseColorTexture); resolve(m.pbrMetallicRoughness.metallicRoughnessTexture); }
    return m;
  });
  var rootLights = json.extensions && json.extensions.KHR_lights_punctual ? json.extensions.KHR_lights_punctual.lights : [];
  var meshes = (json.meshes || []).map(function (mesh) {
    return {
      name: mesh.name,
      primitives: mesh.primitives.map(function (p) {
        var attributes = {};
        Object.keys(p.attributes).forEach(function (k) { attributes[k] = readAccessor(json, bin, p.attributes[k]); });
        return {
          attributes: attributes,
          indices: p.indices !== undefined ? readAccessor(json, bin, p.indices) : undefined,
          material: p.material !== undefined ? materials[p.material] : undefined,
          mode: p.mode === undefined ? 4 : p.mode,
        };
      }),
    };
  });
  var nodes = (json.nodes || []).map(function (n) {
    var o = { name: n.name, matrix: n.matrix, translation: n.translation, rotation: n.rotation, scale: n.scale };
    if (n.mesh !== undefined) o.mesh = meshes[n.mesh];
    if (n.extensions && n.extensions.KHR_lights_punctual) o.light = n.extensions.KHR_lights_punctual.light;
    return o;
  });
  (json.nodes || []).forEach(function (n, i) {
    if (n.children) nodes[i].children = n.children.map(function (c) { return nodes[c]; });
  });
  return { nodes: nodes, meshes: meshes, materials: materials, textures: textures, images: images, lights: rootLights, json: json };
}

module.exports = { parseGLB: parseGLB, readAccessor: readAccessor, loadGLB: loadGLB };
