#!/usr/bin/env node
'use strict';
/** prepare_cli.js <model.glb> <outDir>: runs the host's glTF -> SceneData path (gltf.js + scene_prep.js, i.e.
 *  loader.ts + gpu.ts + bvh.ts of the reference) and writes triangles/materials/bvhNodes/lights .bin blobs
 *  plus info.json. No GPU needed. */
var fs = require('fs'), path = require('path');
var s = require('./scene_prep').prepareScene(require('./gltf').loadGLB(process.argv[2]));
var dir = process.argv[3];
Object.keys(s.blobs).forEach(function (k) { fs.writeFileSync(path.join(dir, k + '.bin'), Buffer.from(s.blobs[k])); });
fs.writeFileSync(path.join(dir, 'info.json'), JSON.stringify({ counts: s.counts, bvhDepth: s.bvhDepth }));
console.log(JSON.stringify(s.counts));
