'use strict';
/**
 * scene_file.js — reads a .ptscene container (written by ptmi/scene_io.py): the byte blobs of the
 * compute pass's storage buffers plus the optional atlas, ready for uploadScene / uploadAtlas.
 *   "PTSC" | u32 version | u32 jsonLength | json | blobs (each 16-byte aligned from file start)
 * json: { triangles|materials|bvhNodes|lights: {offset, length}, atlas?: {offset, length, width, height, format} }
 */
var fs = require('fs');

function slice(buf, e) {
  return buf.buffer.slice(buf.byteOffset + e.offset, buf.byteOffset + e.offset + e.length);
}

function readSceneFile(path) {
  var buf = fs.readFileSync(path);
  if (buf.toString('latin1', 0, 4) !== 'PTSC') throw new Error(path + ': not a .ptscene file');
  var version = buf.readUInt32LE(4), jsonLen = buf.readUInt32LE(8);
  if (version !== 1) throw new Error(path + ': unsupported version ' + version);
  var meta = JSON.parse(buf.toString('utf8', 12, 12 + jsonLen));
  var out = {
    blobs: {
      triangles: slice(buf, meta.triangles), materials: slice(buf, meta.materials),
      bvhNodes: slice(buf, meta.bvhNodes), lights: slice(buf, meta.lights),
    },
    atlas: null, meta: meta,
  };
  if (meta.atlas) {
    out.atlas = { data: slice(buf, meta.atlas), width: meta.atlas.width, height: meta.atlas.height, format: meta.atlas.format };
  }
  return out;
}

module.exports = { readSceneFile: readSceneFile };
