#!/usr/bin/env node
'use strict';
/** pack_cli.js <sceneData.json> <outDir>: packs a SceneData JSON with pack.js and writes the four blobs
 *  + a camera blob (used by tests to compare the JS packing with the numpy layouts byte for byte). */
var fs = require('fs'), path = require('path'), pack = require('./pack');
var data = JSON.parse(fs.readFileSync(process.argv[2], 'utf8')), dir = process.argv[3];
var blobs = pack.packScene(data.scene);
Object.keys(blobs).forEach(function (k) { fs.writeFileSync(path.join(dir, k + '.bin'), Buffer.from(blobs[k])); });
fs.writeFileSync(path.join(dir, 'camera.bin'), Buffer.from(pack.packCamera(data.camera)));
console.log('ok');
