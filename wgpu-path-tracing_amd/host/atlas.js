'use strict';
/**
 * atlas.js — texture atlas of the compute pass's binding 6, following src/renderer/atlas.ts:
 *   packing()      :33-75   four boxes per material (normal, albedo, pbr, emissive — also the EMPTY ones, the
 *                           `if (box)` tests of :56-59 are always true), potpack, power-of-two square size
 *   toBox()        :77-95   box = image size * TEXTURE_PIXEL_RATIO (0.5); fractional for odd sizes
 *   buildCanvas()  :97-184  black opaque canvas; albedo maps pass through an 8-bit sRGB->linear step
 *                           (pow(c/255, 2.2)*255 stored to a Uint8ClampedArray, :138-142); the others are only scaled
 *   upload                  renderer.ts:246-261 copies the canvas into an rgba16float texture: texel = f16(c / 255)
 *
 * Rectangle placement is potpack@2.0.0's published algorithm (the version pinned by the reference's lock file),
 * restated here: boxes sorted by height, descending (stable); a list of free spaces starting with one strip of width
 * max(ceil(sqrt(area / 0.95)), widest box) and unbounded height; each box goes to the top-left corner of the LAST
 * listed space that holds it, and the space is shrunk, removed or split in two.
 *
 * The browser's drawImage resampling is implementation-defined, so texel VALUES of scaled images are
 * "parity unpinned" (DESIGN.md §9). This build defines them as: a canvas pixel is written when its centre lies inside
 * the target rectangle (x, y, w, h); it samples the premultiplied source bilinearly at the source position of that
 * centre, ((px + 0.5 - x) * sw / w - 0.5, (py + 0.5 - y) * sh / h - 0.5), with edge clamp — for the reference's fixed
 * 0.5 ratio and even image sizes that is the mean of a 2x2 block — rounded half up to 8 bits. Placement, rectangle values,
 * atlas size, the gamma step and the f16 conversion are exact restatements.
 */
var TEXTURE_PIXEL_RATIO = 0.5;      // atlas.ts:10

/** potpack@2.0.0: sets x, y on every box (and reorders the array), returns the bounding size */
function potpack(boxes) {
  var area = 0, widest = 0, i;
  for (i = 0; i < boxes.length; i++) { area += boxes[i].w * boxes[i].h; widest = Math.max(widest, boxes[i].w); }
  boxes.sort(function (p, q) { return q.h - p.h; });
  var free = [{ x: 0, y: 0, w: Math.max(Math.ceil(Math.sqrt(area / 0.95)), widest), h: Infinity }];
  var W = 0, H = 0;
  for (i = 0; i < boxes.length; i++) {
    var box = boxes[i];
    for (var k = free.length - 1; k >= 0; k--) {
      var sp = free[k];
      if (box.w > sp.w || box.h > sp.h) continue;
      box.x = sp.x; box.y = sp.y;
      H = Math.max(H, box.y + box.h); W = Math.max(W, box.x + box.w);
      var sameW = box.w === sp.w, sameH = box.h === sp.h;
      if (sameW && sameH) {                       // exact fit: drop the space (the last one takes its slot)
        var tail = free.pop();
        if (k < free.length) free[k] = tail;
      } else if (sameH) { sp.x += box.w; sp.w -= box.w; } else if (sameW) { sp.y += box.h; sp.h -= box.h; } else {
        free.push({ x: sp.x + box.w, y: sp.y, w: sp.w - box.w, h: box.h });   // right of the box, box-high
        sp.y += box.h; sp.h -= box.h;                                            // below the box, full width
      }
      break;
    }
  }
  return { w: W, h: H, fill: (area / (W * H)) || 0 };
}

function imageOf(texInfo) {
  var t = texInfo && texInfo.texture;
  return (t && t.source && t.source.image) || null;
}

/** atlas.ts:77-95 */
function toBox(texInfo) {
  var img = imageOf(texInfo);
  if (!img) return { w: 0, h: 0, x: 0, y: 0 };
  return { w: (img.width || 0) * TEXTURE_PIXEL_RATIO, h: (img.height || 0) * TEXTURE_PIXEL_RATIO, x: 0, y: 0 };
}

/** destination pixels whose centres lie inside [start, start + extent): first index and count */
function pixelSpan(start, extent) {
  if (!(extent > 0)) return { first: 0, count: 0 };
  var first = Math.ceil(start - 0.5), end = Math.ceil(start + extent - 0.5);
  return { first: first, count: Math.max(0, end - first) };
}

/** img (RGBA8, straight alpha) scaled into the rectangle (dx, dy, dw, dh): premultiplied float rgb + alpha per covered pixel */
function resample(img, dx, dy, dw, dh) {
  var cols = pixelSpan(dx, dw), rows = pixelSpan(dy, dh), nw = cols.count, nh = rows.count;
  var sw = img.width, sh = img.height, src = img.data;
  var out = new Float64Array(nw * nh * 4);
  function texel(x, y, c) {
    var o = (y * sw + x) * 4;
    return c === 3 ? src[o + 3] : src[o + c] * src[o + 3] / 255;
  }
  for (var j = 0; j < nh; j++) {
    var fy = (rows.first + j + 0.5 - dy) * sh / dh - 0.5, y0 = Math.floor(fy), ty = fy - y0;
    var ya = Math.min(Math.max(y0, 0), sh - 1), yb = Math.min(Math.max(y0 + 1, 0), sh - 1);
    for (var i = 0; i < nw; i++) {
      var fx = (cols.first + i + 0.5 - dx) * sw / dw - 0.5, x0 = Math.floor(fx), tx = fx - x0;
      var xa = Math.min(Math.max(x0, 0), sw - 1), xb = Math.min(Math.max(x0 + 1, 0), sw - 1);
      for (var c = 0; c < 4; c++) {
        var top = (1 - tx) * texel(xa, ya, c) + tx * texel(xb, ya, c), bot = (1 - tx) * texel(xa, yb, c) + tx * texel(xb, yb, c);
        out[(j * nw + i) * 4 + c] = (1 - ty) * top + ty * bot;
      }
    }
  }
  return { x: cols.first, y: rows.first, width: nw, height: nh, data: out };
}

function halfUp(v) { return Math.min(255, Math.max(0, Math.floor(v + 0.5))); }

/** atlas.ts:97-184: the size x size RGBA8 canvas */
function buildCanvas(size, entries) {
  var canvas = new Uint8Array(size * size * 4);
  for (var p = 0; p < size * size; p++) canvas[p * 4 + 3] = 255;                // fillStyle 'black'
  function draw(info, texInfo, isAlbedo) {
    var img = imageOf(texInfo);
    if (!img) return;
    var s = resample(img, info.x, info.y, info.w, info.h), px = new Uint8ClampedArray(4);
    for (var j = 0; j < s.height; j++) {
      for (var i = 0; i < s.width; i++) {
        var o = (j * s.width + i) * 4, a = s.data[o + 3], r = s.data[o], g = s.data[o + 1], b = s.data[o + 2];
        if (isAlbedo) {
          // temp canvas (premultiplied 8-bit) -> getImageData (straight) -> gamma -> putImageData -> drawImage
          var a8 = halfUp(a);
          var un = function (v) { return a8 === 0 ? 0 : halfUp(halfUp(v) * 255 / a8); };
          px[0] = Math.pow(un(r) / 255, 2.2) * 255; px[1] = Math.pow(un(g) / 255, 2.2) * 255; px[2] = Math.pow(un(b) / 255, 2.2) * 255;
          r = px[0] * a8 / 255; g = px[1] * a8 / 255; b = px[2] * a8 / 255;
        }
        var x = s.x + i, y = s.y + j;
        if (x < 0 || y < 0 || x >= size || y >= size) continue;
        var d = (y * size + x) * 4;                                             // source-over on opaque black
        canvas[d] = halfUp(r); canvas[d + 1] = halfUp(g); canvas[d + 2] = halfUp(b);
      }
    }
  }
  entries.forEach(function (e) {
    var m = e.material, pbr = m.pbrMetallicRoughness || {};
    draw(e.textures.albedoMap, pbr.baseColorTexture, true);
    draw(e.textures.normalMap, m.normalTexture, false);
    draw(e.textures.pbrMap, pbr.metallicRoughnessTexture, false);
    draw(e.textures.emissiveMap, m.emissiveTexture, false);
  });
  return canvas;
}

/** f32 -> IEEE binary16, round to nearest even */
var f32buf = new Float32Array(1), u32buf = new Uint32Array(f32buf.buffer);
function toHalf(v) {
  f32buf[0] = v;
  var x = u32buf[0], sign = (x >>> 16) & 0x8000, exp = (x >>> 23) & 0xff, man = x & 0x7fffff;
  if (exp === 0xff) return sign | 0x7c00 | (man ? 0x200 : 0);
  var e = exp - 127 + 15;
  if (e >= 31) return sign | 0x7c00;
  if (e <= 0) {
    if (e < -10) return sign;
    man |= 0x800000;
    var shift = 14 - e, h = man >>> shift, rem = man & ((1 << shift) - 1), half = 1 << (shift - 1);
    if (rem > half || (rem === half && (h & 1))) h++;
    return sign | h;
  }
  var out = sign | (e << 10) | (man >>> 13), r = man & 0x1fff;
  if (r > 0x1000 || (r === 0x1000 && (out & 1))) out++;
  return out;
}

var unormHalf = null;
/** RGBA8 canvas -> rgba16float texels (renderer.ts:246-261) */
function canvasToHalf(rgba8) {
  if (!unormHalf) { unormHalf = new Uint16Array(256); for (var c = 0; c < 256; c++) unormHalf[c] = toHalf(Math.fround(c / 255)); }
  var out = new Uint16Array(rgba8.length);
  for (var i = 0; i < rgba8.length; i++) out[i] = unormHalf[rgba8[i]];
  return out;
}

/**
 * atlas.ts:33-75. gltf.materials carry resolved texture references ({texture: {source: {image}}}, as after
 * postProcessGLTF). Returns { texture: {width, height, rgba8, data (f16 bits), format: 1}, materials: Map }.
 */
function packing(gltf) {
  var boxes = [], materials = new Map(), entries = [];
  (gltf.materials || []).forEach(function (m) {
    var pbr = m.pbrMetallicRoughness || {};
    var normalBox = toBox(m.normalTexture), albedoBox = toBox(pbr.baseColorTexture);
    var pbrBox = toBox(pbr.metallicRoughnessTexture), emissionBox = toBox(m.emissiveTexture);
    var textures = { albedoMap: albedoBox, normalMap: normalBox, pbrMap: pbrBox, emissiveMap: emissionBox };
    materials.set(m, textures);
    entries.push({ material: m, textures: textures });
    boxes.push(normalBox, albedoBox, pbrBox, emissionBox);
  });
  var packed = potpack(boxes);
  var size = Math.max(1, Math.pow(2, Math.ceil(Math.log2(Math.max(packed.w, packed.h)))));
  var rgba8 = buildCanvas(size, entries);
  return {
    texture: { width: size, height: size, rgba8: rgba8, data: canvasToHalf(rgba8), format: 1 },
    materials: materials, packed: packed,
  };
}

module.exports = { potpack: potpack, packing: packing, toBox: toBox, toHalf: toHalf, canvasToHalf: canvasToHalf,
  resample: resample, TEXTURE_PIXEL_RATIO: TEXTURE_PIXEL_RATIO };
