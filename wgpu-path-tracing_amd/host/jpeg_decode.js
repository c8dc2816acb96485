'use strict';
/**
 * jpeg_decode.js — JPEG -> RGBA8, for glTF textures
 * (`texture.source.image` of src/renderer/atlas.ts:76-95, which the reference gets from the browser).
 *
 * Browsers decode JPEG with libjpeg-turbo's defaults, so this follows the IJG library's published algorithms
 * step for step and reproduces its output exactly: the accurate integer inverse DCT (jidctint.c, 13-bit constants,
 * two passes with DESCALE rounding), "fancy" chroma upsampling for 2:1 horizontal and 2:1 x 2:1 sampling
 * (jdsample.c triangle filters; other factors replicate), and the fixed-point YCbCr -> RGB tables of jdcolor.c.
 * Sequential and progressive Huffman files (jdhuff.c, jdphuff.c), 8-bit samples, 1 (grey) or 3 (YCbCr) components,
 * restart intervals. Arithmetic-coded, lossless, 12-bit and CMYK files are rejected with an error.
 */

var ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63];

/** canonical Huffman table -> lookup by (length, code) */
function buildHuffman(counts, symbols) {
  var table = { maxcode: new Int32Array(18), valptr: new Int32Array(17), mincode: new Int32Array(17), symbols: symbols };
  var code = 0, k = 0;
  for (var len = 1; len <= 16; len++) {
    table.valptr[len] = k; table.mincode[len] = code;
    code += counts[len - 1]; k += counts[len - 1];
    table.maxcode[len] = counts[len - 1] ? code - 1 : -1;
    code <<= 1;
  }
  table.maxcode[17] = 0x7fffffff;
  return table;
}

function BitReader(data, pos) { this.data = data; this.pos = pos; this.acc = 0; this.bits = 0; this.marker = 0; }
BitReader.prototype.bit = function () {
  if (this.bits === 0) {
    var b = 0;
    if (!this.marker && this.pos < this.data.length) {
      b = this.data[this.pos++];
      if (b === 0xff) {
        var n = this.data[this.pos];
        if (n === 0) this.pos++;                         // stuffed zero
        else { this.marker = n; this.pos--; b = 0; }      // a marker: feed zeros until the caller handles it
      }
    }
    this.acc = b; this.bits = 8;
  }
  this.bits--;
  return (this.acc >> this.bits) & 1;
};
BitReader.prototype.receive = function (n) { var v = 0; while (n-- > 0) v = (v << 1) | this.bit(); return v; };
BitReader.prototype.decode = function (t) {
  var code = 0;
  for (var len = 1; len <= 16; len++) {
    code = (code << 1) | this.bit();
    if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len]) return t.symbols[t.valptr[len] + code - t.mincode[len]];
  }
  throw new Error('JPEG: bad Huffman code');
};
/** after a restart marker: drop the partial byte and step over RSTn */
BitReader.prototype.restart = function () {
  this.bits = 0;
  if (!this.marker) {                                     // marker not reached through the bit buffer yet
    while (this.pos < this.data.length && !(this.data[this.pos] === 0xff && this.data[this.pos + 1] >= 0xd0 && this.data[this.pos + 1] <= 0xd7)) this.pos++;
  }
  this.pos += 2; this.marker = 0;
};

function extend(v, n) { return n === 0 ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); }

var F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
  F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
  F_2_562915447 = 20995, F_3_072711026 = 25172;

/** jidctint.c: coef (64 dequantised Int32, natural order) -> 64 samples 0..255 written to out[o + y*stride + x] */
var ws = new Int32Array(64);
function idct(coef, out, o, stride) {
  var i, p, z1, z2, z3, z4, z5, t0, t1, t2, t3, t10, t11, t12, t13;
  for (i = 0; i < 8; i++) {                               // pass 1: columns, results scaled by 2^PASS1_BITS (2)
    if (!coef[8 + i] && !coef[16 + i] && !coef[24 + i] && !coef[32 + i] && !coef[40 + i] && !coef[48 + i] && !coef[56 + i]) {
      var dc = coef[i] << 2;
      for (p = 0; p < 8; p++) ws[p * 8 + i] = dc;
      continue;
    }
    z2 = coef[16 + i]; z3 = coef[48 + i];
    z1 = (z2 + z3) * F_0_541196100; t2 = z1 - z3 * F_1_847759065; t3 = z1 + z2 * F_0_765366865;
    z2 = coef[i]; z3 = coef[32 + i];
    t0 = (z2 + z3) << 13; t1 = (z2 - z3) << 13;
    t10 = t0 + t3; t13 = t0 - t3; t11 = t1 + t2; t12 = t1 - t2;
    t0 = coef[56 + i]; t1 = coef[40 + i]; t2 = coef[24 + i]; t3 = coef[8 + i];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3; z5 = (z3 + z4) * F_1_175875602;
    t0 *= F_0_298631336; t1 *= F_2_053119869; t2 *= F_3_072711026; t3 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    ws[i] = (t10 + t3 + 1024) >> 11; ws[56 + i] = (t10 - t3 + 1024) >> 11;
    ws[8 + i] = (t11 + t2 + 1024) >> 11; ws[48 + i] = (t11 - t2 + 1024) >> 11;
    ws[16 + i] = (t12 + t1 + 1024) >> 11; ws[40 + i] = (t12 - t1 + 1024) >> 11;
    ws[24 + i] = (t13 + t0 + 1024) >> 11; ws[32 + i] = (t13 - t0 + 1024) >> 11;
  }
  function put(at, v) { v = (v >> 18) + 128; out[at] = v < 0 ? 0 : v > 255 ? 255 : v; }   // DESCALE by 13 + 2 + 3, level shift, clamp
  for (i = 0; i < 8; i++) {                               // pass 2: rows
    p = i * 8;
    var r = o + i * stride;
    z2 = ws[p + 2]; z3 = ws[p + 6];
    z1 = (z2 + z3) * F_0_541196100; t2 = z1 - z3 * F_1_847759065; t3 = z1 + z2 * F_0_765366865;
    t0 = (ws[p] + ws[p + 4]) << 13; t1 = (ws[p] - ws[p + 4]) << 13;
    t10 = t0 + t3; t13 = t0 - t3; t11 = t1 + t2; t12 = t1 - t2;
    t0 = ws[p + 7]; t1 = ws[p + 5]; t2 = ws[p + 3]; t3 = ws[p + 1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3; z5 = (z3 + z4) * F_1_175875602;
    t0 *= F_0_298631336; t1 *= F_2_053119869; t2 *= F_3_072711026; t3 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    var h = 1 << 17;
    put(r, t10 + t3 + h); put(r + 7, t10 - t3 + h); put(r + 1, t11 + t2 + h); put(r + 6, t11 - t2 + h);
    put(r + 2, t12 + t1 + h); put(r + 5, t12 - t1 + h); put(r + 3, t13 + t0 + h); put(r + 4, t13 - t0 + h);
  }
}

/**
 * jdsample.c: component plane -> full-resolution plane. The plane holds whole 8x8 blocks; like the library, the
 * filters read the decoded padding COLUMN next to a one-column component, but below the last real row
 * (downsampled height) they see that row again (jdmainct.c set_bottom_pointers), and above the first the first.
 */
function upsample(comp, hmax, vmax, width, height) {
  var hs = hmax / comp.h, vs = vmax / comp.v, src = comp.plane, stride = comp.stride;
  var cw = Math.ceil(width * comp.h / hmax), ch = Math.ceil(height * comp.v / vmax);       // downsampled size
  var out = new Uint8Array(width * height), x, y, row;
  if (hs === 1 && vs === 1) {
    for (y = 0; y < height; y++) for (x = 0; x < width; x++) out[y * width + x] = src[y * stride + x];
  } else if (hs === 2 && vs === 1) {                       // h2v1_fancy_upsample
    for (y = 0; y < height; y++) {
      row = y * stride;
      for (x = 0; x < width; x++) {
        var i = x >> 1, c = src[row + i], v;
        if (x & 1) v = (i === cw - 1 && cw > 1) ? c : (c * 3 + src[row + i + 1] + 2) >> 2;
        else v = i === 0 ? c : (c * 3 + src[row + i - 1] + 1) >> 2;
        out[y * width + x] = v;
      }
    }
  } else if (hs === 2 && vs === 2) {                       // h2v2_fancy_upsample
    for (y = 0; y < height; y++) {
      var near = (y >> 1) * stride, fr = (y & 1) ? (y >> 1) + 1 : (y >> 1) - 1;
      var far = Math.min(Math.max(fr, 0), ch - 1) * stride;
      for (x = 0; x < width; x++) {
        var j = x >> 1, cur = src[near + j] * 3 + src[far + j], o2;
        if (x & 1) o2 = (j === cw - 1 && cw > 1) ? (cur * 4 + 7) >> 4 : (cur * 3 + src[near + j + 1] * 3 + src[far + j + 1] + 7) >> 4;
        else o2 = j === 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + src[near + j - 1] * 3 + src[far + j - 1] + 8) >> 4;
        out[y * width + x] = o2;
      }
    }
  } else {                                                 // int_upsample: replication
    if (hs !== Math.floor(hs) || vs !== Math.floor(vs)) throw new Error('JPEG: fractional sampling ratios are not supported');
    for (y = 0; y < height; y++) for (x = 0; x < width; x++) out[y * width + x] = src[Math.floor(y / vs) * stride + Math.floor(x / hs)];
  }
  return out;
}

function decodeJPEG(buf) {
  if (buf[0] !== 0xff || buf[1] !== 0xd8) throw new Error('not a JPEG file');
  var pos = 2, qt = [], dcT = [], acT = [], frame = null, restartInterval = 0, adobeTransform = -1;
  function u16(at) { return (buf[at] << 8) | buf[at + 1]; }
  while (pos < buf.length) {
    if (buf[pos] !== 0xff) { pos++; continue; }
    var marker = buf[pos + 1];
    pos += 2;
    if (marker === 0xff) { pos--; continue; }
    if (marker === 0xd8 || marker === 0x01 || (marker >= 0xd0 && marker <= 0xd7)) continue;
    if (marker === 0xd9) break;
    var len = u16(pos), seg = pos + 2, end = pos + len;
    if (marker === 0xdb) {                                 // DQT
      while (seg < end) {
        var pq = buf[seg] >> 4, tq = buf[seg] & 15; seg++;
        var q = new Int32Array(64);
        for (var k = 0; k < 64; k++) { q[k] = pq ? u16(seg) : buf[seg]; seg += pq ? 2 : 1; }   // zigzag order, like the coefficients
        qt[tq] = q;
      }
    } else if (marker === 0xc4) {                          // DHT
      while (seg < end) {
        var tc = buf[seg] >> 4, th = buf[seg] & 15; seg++;
        var counts = buf.slice(seg, seg + 16), total = 0;
        for (var c = 0; c < 16; c++) total += counts[c];
        seg += 16;
        (tc ? acT : dcT)[th] = buildHuffman(counts, buf.slice(seg, seg + total));
        seg += total;
      }
    } else if (marker === 0xc0 || marker === 0xc1 || marker === 0xc2) {   // SOF0 / SOF1 / SOF2
      if (frame) throw new Error('JPEG: more than one frame');
      if (buf[seg] !== 8) throw new Error('JPEG: only 8-bit samples are supported');
      frame = { height: u16(seg + 1), width: u16(seg + 3), comps: [], progressive: marker === 0xc2 };
      var nc = buf[seg + 5];
      if (nc !== 1 && nc !== 3) throw new Error('JPEG: ' + nc + '-component images are not supported');
      for (var ci = 0; ci < nc; ci++) {
        var b = seg + 6 + ci * 3;
        frame.comps.push({ id: buf[b], h: buf[b + 1] >> 4, v: buf[b + 1] & 15, tq: buf[b + 2] });
      }
      layoutFrame(frame);
    } else if (marker >= 0xc3 && marker <= 0xcf && marker !== 0xc4 && marker !== 0xc8 && marker !== 0xcc) {
      throw new Error('JPEG: unsupported coding process (SOF' + (marker - 0xc0) + ')');
    } else if (marker === 0xdd) { restartInterval = u16(seg);
    } else if (marker === 0xee && len >= 14 && buf.toString('latin1', seg, seg + 5) === 'Adobe') { adobeTransform = buf[seg + 11];
    } else if (marker === 0xda) {                          // SOS
      if (!frame) throw new Error('JPEG: scan before frame header');
      var ns = buf[seg], scan = [];
      for (var s = 0; s < ns; s++) {
        var cid = buf[seg + 1 + s * 2], tbl = buf[seg + 2 + s * 2];
        var comp = frame.comps.filter(function (cc) { return cc.id === cid; })[0];
        if (!comp) throw new Error('JPEG: scan names an unknown component');
        comp.dc = dcT[tbl >> 4]; comp.ac = acT[tbl & 15];
        scan.push(comp);
      }
      var ss = buf[seg + 1 + ns * 2], se = buf[seg + 2 + ns * 2], ah = buf[seg + 3 + ns * 2] >> 4, al = buf[seg + 3 + ns * 2] & 15;
      pos = decodeScan(buf, end, frame, scan, restartInterval, ss, se, ah, al);
      continue;
    }
    pos = end;
  }
  if (!frame || !frame.scans) throw new Error('JPEG: no image data');
  var w = frame.width, h = frame.height, out = new Uint8Array(w * h * 4), i;
  frame.comps.forEach(function (c2) {                      // dequantise + inverse DCT of every block
    var q = qt[c2.tq], block = new Int32Array(64);
    if (!q) throw new Error('JPEG: missing quantisation table');
    for (var by = 0; by < c2.blocksY; by++) {
      for (var bx = 0; bx < c2.blocksX; bx++) {
        var base = (by * c2.blocksX + bx) * 64;
        for (var k = 0; k < 64; k++) block[ZIGZAG[k]] = c2.coef[base + k] * q[k];
        idct(block, c2.plane, by * 8 * c2.stride + bx * 8, c2.stride);
      }
    }
  });
  var planes = frame.comps.map(function (c2) { return upsample(c2, frame.hmax, frame.vmax, w, h); });
  if (planes.length === 1) {
    for (i = 0; i < w * h; i++) { out[i * 4] = out[i * 4 + 1] = out[i * 4 + 2] = planes[0][i]; out[i * 4 + 3] = 255; }
  } else if (adobeTransform === 0) {                       // Adobe marker says the components already are RGB
    for (i = 0; i < w * h; i++) { out[i * 4] = planes[0][i]; out[i * 4 + 1] = planes[1][i]; out[i * 4 + 2] = planes[2][i]; out[i * 4 + 3] = 255; }
  } else {                                                 // jdcolor.c ycc_rgb_convert, 16-bit fixed point
    var crR = new Int32Array(256), cbB = new Int32Array(256), crG = new Int32Array(256), cbG = new Int32Array(256);
    for (i = 0; i < 256; i++) {
      var x = i - 128;
      crR[i] = (91881 * x + 32768) >> 16; cbB[i] = (116130 * x + 32768) >> 16;
      crG[i] = -46802 * x; cbG[i] = -22554 * x + 32768;
    }
    var clamp = function (v) { return v < 0 ? 0 : v > 255 ? 255 : v; };
    for (i = 0; i < w * h; i++) {
      var Y = planes[0][i], cb = planes[1][i], cr = planes[2][i];
      out[i * 4] = clamp(Y + crR[cr]); out[i * 4 + 1] = clamp(Y + ((cbG[cb] + crG[cr]) >> 16));
      out[i * 4 + 2] = clamp(Y + cbB[cb]); out[i * 4 + 3] = 255;
    }
  }
  return { width: w, height: h, data: out };
}

/** block grid of every component: padded to whole MCUs; a one-component frame has 8x8 MCUs whatever its factors */
function layoutFrame(frame) {
  var hmax = 1, vmax = 1;
  if (frame.comps.length === 1) { frame.comps[0].h = 1; frame.comps[0].v = 1; }
  frame.comps.forEach(function (c) { hmax = Math.max(hmax, c.h); vmax = Math.max(vmax, c.v); });
  frame.hmax = hmax; frame.vmax = vmax;
  frame.mcusX = Math.ceil(frame.width / (8 * hmax)); frame.mcusY = Math.ceil(frame.height / (8 * vmax));
  frame.comps.forEach(function (c) {
    c.blocksX = frame.mcusX * c.h; c.blocksY = frame.mcusY * c.v;
    c.ownX = Math.ceil(Math.ceil(frame.width * c.h / hmax) / 8);      // blocks a non-interleaved scan of it covers
    c.ownY = Math.ceil(Math.ceil(frame.height * c.v / vmax) / 8);
    c.stride = c.blocksX * 8;
    c.plane = new Uint8Array(c.stride * c.blocksY * 8);
    c.coef = new Int16Array(c.blocksX * c.blocksY * 64);              // zigzag order per block
  });
}

/** one scan (jdhuff.c / jdphuff.c); returns the position after its entropy-coded data */
function decodeScan(buf, pos, frame, scan, restartInterval, ss, se, ah, al) {
  var br = new BitReader(buf, pos), eobrun = 0, prog = frame.progressive;
  if (!prog) { ss = 0; se = 63; ah = 0; al = 0; }
  scan.forEach(function (c) {
    c.pred = 0;
    if ((ss === 0 && !c.dc && !(prog && ah)) || (se > 0 && !c.ac)) throw new Error('JPEG: missing Huffman table');
  });
  if (prog && ss > 0 && scan.length !== 1) throw new Error('JPEG: interleaved AC scan');

  function block(c, at) {
    var co = c.coef, k, rs, r, s, z;
    if (!prog) {                                           // sequential: DC difference then run/size pairs
      s = br.decode(c.dc); c.pred += extend(br.receive(s), s); co[at] = c.pred;
      for (k = 1; k < 64;) {
        rs = br.decode(c.ac); r = rs >> 4; s = rs & 15;
        if (s === 0) { if (r === 15) { k += 16; continue; } break; }
        k += r;
        if (k > 63) throw new Error('JPEG: coefficient index out of range');
        co[at + k] = extend(br.receive(s), s); k++;
      }
    } else if (ss === 0) {                                 // DC scan: first pass or one more bit
      if (ah === 0) { s = br.decode(c.dc); c.pred += extend(br.receive(s), s); co[at] = c.pred * (1 << al); }
      else if (br.bit()) co[at] |= 1 << al;
    } else if (ah === 0) {                                 // AC first pass with end-of-band runs
      if (eobrun > 0) { eobrun--; return; }
      for (k = ss; k <= se;) {
        rs = br.decode(c.ac); r = rs >> 4; s = rs & 15;
        if (s === 0) {
          if (r < 15) { eobrun = br.receive(r) + (1 << r) - 1; break; }
          k += 16; continue;
        }
        k += r;
        if (k > 63) throw new Error('JPEG: coefficient index out of range');
        co[at + k] = extend(br.receive(s), s) * (1 << al); k++;
      }
    } else {                                               // AC refinement (jdphuff.c decode_mcu_AC_refine)
      var p1 = 1 << al, m1 = -1 << al;
      k = ss;
      if (eobrun <= 0) {
        for (; k <= se; k++) {
          rs = br.decode(c.ac); r = rs >> 4; s = rs & 15;
          if (s) s = br.bit() ? p1 : m1;
          else if (r !== 15) { eobrun = 1 << r; if (r) eobrun += br.receive(r); break; }
          do {
            z = co[at + k];
            if (z !== 0) { if (br.bit() && (z & p1) === 0) co[at + k] = z + (z >= 0 ? p1 : m1); }
            else if (--r < 0) break;
            k++;
          } while (k <= se);
          if (s && k <= 63) co[at + k] = s;
        }
      }
      if (eobrun > 0) {
        for (; k <= se; k++) {
          z = co[at + k];
          if (z !== 0 && br.bit() && (z & p1) === 0) co[at + k] = z + (z >= 0 ? p1 : m1);
        }
        eobrun--;
      }
    }
  }

  var count = 0;
  function restartCheck() {
    if (restartInterval && count > 0 && count % restartInterval === 0) {
      br.restart(); eobrun = 0;
      scan.forEach(function (c) { c.pred = 0; });
    }
    count++;
  }
  if (scan.length === 1) {                                 // non-interleaved: the component's own block rows
    var c = scan[0];
    for (var by = 0; by < c.ownY; by++) for (var bx = 0; bx < c.ownX; bx++) { restartCheck(); block(c, (by * c.blocksX + bx) * 64); }
  } else {
    for (var my = 0; my < frame.mcusY; my++) {
      for (var mx = 0; mx < frame.mcusX; mx++) {
        restartCheck();
        for (var ci = 0; ci < scan.length; ci++) {
          var cc = scan[ci];
          for (var v = 0; v < cc.v; v++) for (var hh = 0; hh < cc.h; hh++) block(cc, ((my * cc.v + v) * cc.blocksX + mx * cc.h + hh) * 64);
        }
      }
    }
  }
  frame.scans = (frame.scans || 0) + 1;
  // the next marker: where the bit reader stopped, or the first one after its position
  var p = br.pos;
  while (p < buf.length - 1 && !(buf[p] === 0xff && buf[p + 1] !== 0 && !(buf[p + 1] >= 0xd0 && buf[p + 1] <= 0xd7))) p++;
  return p;
}

module.exports = { decodeJPEG: decodeJPEG };
