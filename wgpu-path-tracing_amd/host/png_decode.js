'use strict';
/**
 * png_decode.js — PNG -> RGBA8, what the reference gets from the browser's createImageBitmap for the
 * images @loaders.gl/gltf hands to src/renderer/atlas.ts:76-95 (`texture.source.image`).
 * Non-interlaced PNG of every colour type and bit depth (16-bit samples keep their high byte), with tRNS.
 * Gamma / ICC chunks are ignored (browsers do not colour-manage untagged 8-bit PNGs either).
 */
var zlib = require('zlib');

var SIGNATURE = [137, 80, 78, 71, 13, 10, 26, 10];
var CHANNELS = { 0: 1, 2: 3, 3: 1, 4: 2, 6: 4 };

function paeth(a, b, c) {
  var p = a + b - c, pa = Math.abs(p - a), pb = Math.abs(p - b), pc = Math.abs(p - c);
  return pa <= pb && pa <= pc ? a : pb <= pc ? b : c;
}

/** undo the per-row filters in place; returns the rows without their filter bytes */
function unfilter(raw, rowBytes, height, bpp) {
  var out = Buffer.alloc(rowBytes * height);
  for (var y = 0; y < height; y++) {
    var type = raw[y * (rowBytes + 1)], src = y * (rowBytes + 1) + 1, dst = y * rowBytes, up = dst - rowBytes;
    for (var i = 0; i < rowBytes; i++) {
      var a = i >= bpp ? out[dst + i - bpp] : 0, b = y > 0 ? out[up + i] : 0, c = i >= bpp && y > 0 ? out[up + i - bpp] : 0;
      var x = raw[src + i];
      if (type === 1) x += a;
      else if (type === 2) x += b;
      else if (type === 3) x += (a + b) >> 1;
      else if (type === 4) x += paeth(a, b, c);
      else if (type !== 0) throw new Error('PNG: bad filter type ' + type);
      out[dst + i] = x & 255;
    }
  }
  return out;
}

function decodePNG(buf) {
  for (var s = 0; s < 8; s++) if (buf[s] !== SIGNATURE[s]) throw new Error('not a PNG file');
  var off = 8, ihdr = null, palette = null, trns = null, idat = [];
  while (off + 12 <= buf.length) {
    var len = buf.readUInt32BE(off), type = buf.toString('latin1', off + 4, off + 8), data = buf.slice(off + 8, off + 8 + len);
    if (type === 'IHDR') ihdr = data;
    else if (type === 'PLTE') palette = data;
    else if (type === 'tRNS') trns = data;
    else if (type === 'IDAT') idat.push(data);
    else if (type === 'IEND') break;
    off += 12 + len;
  }
  if (!ihdr || !idat.length) throw new Error('PNG: missing IHDR or IDAT');
  var width = ihdr.readUInt32BE(0), height = ihdr.readUInt32BE(4), depth = ihdr[8], ctype = ihdr[9];
  if (ihdr[12] !== 0) throw new Error('PNG: interlaced images are not supported');
  var ch = CHANNELS[ctype];
  if (!ch || [1, 2, 4, 8, 16].indexOf(depth) < 0) throw new Error('PNG: unsupported colour type / bit depth');
  if (ctype === 3 && !palette) throw new Error('PNG: palette image without PLTE');
  var bitsPerPixel = ch * depth, rowBytes = (width * bitsPerPixel + 7) >> 3, bpp = Math.max(1, bitsPerPixel >> 3);
  var raw = zlib.inflateSync(Buffer.concat(idat));
  if (raw.length < (rowBytes + 1) * height) throw new Error('PNG: truncated image data');
  var rows = unfilter(raw, rowBytes, height, bpp);

  var maxv = (1 << Math.min(depth, 8)) - 1;
  function sample(row, index) {              // index-th sample of a row, reduced to 0..maxv (16-bit: high byte)
    if (depth === 8) return rows[row + index];
    if (depth === 16) return rows[row + index * 2];
    var bit = index * depth, byte = rows[row + (bit >> 3)];
    return (byte >> (8 - depth - (bit & 7))) & maxv;
  }
  function sample16(row, index) { return depth === 16 ? rows.readUInt16BE(row + index * 2) : sample(row, index); }
  var keyG = -1, keyR = -1, keyGn = -1, keyB = -1;
  if (trns && ctype === 0) keyG = trns.readUInt16BE(0);
  if (trns && ctype === 2) { keyR = trns.readUInt16BE(0); keyGn = trns.readUInt16BE(2); keyB = trns.readUInt16BE(4); }

  var out = new Uint8Array(width * height * 4);
  for (var y = 0; y < height; y++) {
    var row = y * rowBytes;
    for (var x = 0; x < width; x++) {
      var o = (y * width + x) * 4, r, g, b, a = 255;
      if (ctype === 3) {
        var idx = sample(row, x);
        if (idx * 3 + 2 >= palette.length) throw new Error('PNG: palette index out of range');
        r = palette[idx * 3]; g = palette[idx * 3 + 1]; b = palette[idx * 3 + 2];
        if (trns && idx < trns.length) a = trns[idx];
      } else if (ctype === 0 || ctype === 4) {
        var v = sample(row, x * ch);
        r = g = b = depth < 8 ? Math.round(v * 255 / maxv) : v;
        if (ctype === 4) a = sample(row, x * ch + 1);
        else if (keyG >= 0 && sample16(row, x) === keyG) a = 0;
      } else {
        r = sample(row, x * ch); g = sample(row, x * ch + 1); b = sample(row, x * ch + 2);
        if (ctype === 6) a = sample(row, x * ch + 3);
        else if (keyR >= 0 && sample16(row, x * 3) === keyR && sample16(row, x * 3 + 1) === keyGn && sample16(row, x * 3 + 2) === keyB) a = 0;
      }
      out[o] = r; out[o + 1] = g; out[o + 2] = b; out[o + 3] = a;
    }
  }
  return { width: width, height: height, data: out };
}

module.exports = { decodePNG: decodePNG };
