#!/usr/bin/env node
'use strict';
/** prepare_cli.js <model.glb> <outDir>: runs the host's glTF -> SceneData path (gltf.js + atlas.js + scene_prep.js,
 *  i.e. loader.ts + atlas.ts + gpu.ts + bvh.ts of the reference) and writes triangles/materials/bvhNodes/lights
 *  .bin blobs, the atlas (atlas.bin = rgba16float texels, atlas_rgba8.bin = the 8-bit canvas) and info.json.
 *  No GPU needed. */
var fs = require('fs'), path = require('path');
var s = require('./scene_prep').prepareScene(require('./gltf').loadGLB(process.argv[2]));
var dir = process.argv[3];
Object.keys(s.blobs).forEach(function (k) { fs.writeFileSync(path.join(dir, k + '.bin'), Buffer.from(s.blobs[k])); });
var a = s.atlas;
fs.writeFileSync(path.join(dir, 'atlas.bin'), Buffer.from(a.data.buffer, a.data.byteOffset, a.data.byteLength));
fs.writeFileSync(path.join(dir, 'atlas_rgba8.bin'), Buffer.from(a.rgba8.buffer, a.rgba8.byteOffset, a.rgba8.byteLength));
fs.writeFileSync(path.join(dir, 'info.json'), JSON.stringify({ counts: s.counts, bvhDepth: s.bvhDepth,
  atlas: { width: a.width, height: a.height, format: a.format } }));
console.log(JSON.stringify(s.counts));
