'use strict';
/**
 * mat.js — the subset of wgpu-matrix@3.4.0 (un-vendored dependency of the reference, MIT) that
 * src/renderer/gpu.ts:88-103, :153-191, :203-273 uses, restated: column-major 4x4 matrices, vectors
 * and quaternions stored in Float32Array (wgpu-matrix's default), arithmetic in JS doubles, one f32
 * rounding per stored element. "Parity unpinned": the library itself is not available offline, so
 * last-bit agreement with it is not verified; tests check the maths against float64 references.
 */
function vec3(x, y, z) { var v = new Float32Array(3); v[0] = x || 0; v[1] = y || 0; v[2] = z || 0; return v; }
function mat4() { return new Float32Array(16); }

function identity() { var m = mat4(); m[0] = m[5] = m[10] = m[15] = 1; return m; }
function create16(a) { var m = mat4(); for (var i = 0; i < 16; i++) m[i] = a[i]; return m; }
function clone(a) { return new Float32Array(a); }

/** a * b (apply b first) */
function mul(a, b, dst) {
  dst = dst || mat4();
  var a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7],
    a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11], a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
  var b00 = b[0], b01 = b[1], b02 = b[2], b03 = b[3], b10 = b[4], b11 = b[5], b12 = b[6], b13 = b[7],
    b20 = b[8], b21 = b[9], b22 = b[10], b23 = b[11], b30 = b[12], b31 = b[13], b32 = b[14], b33 = b[15];
  dst[0] = a00 * b00 + a10 * b01 + a20 * b02 + a30 * b03;
  dst[1] = a01 * b00 + a11 * b01 + a21 * b02 + a31 * b03;
  dst[2] = a02 * b00 + a12 * b01 + a22 * b02 + a32 * b03;
  dst[3] = a03 * b00 + a13 * b01 + a23 * b02 + a33 * b03;
  dst[4] = a00 * b10 + a10 * b11 + a20 * b12 + a30 * b13;
  dst[5] = a01 * b10 + a11 * b11 + a21 * b12 + a31 * b13;
  dst[6] = a02 * b10 + a12 * b11 + a22 * b12 + a32 * b13;
  dst[7] = a03 * b10 + a13 * b11 + a23 * b12 + a33 * b13;
  dst[8] = a00 * b20 + a10 * b21 + a20 * b22 + a30 * b23;
  dst[9] = a01 * b20 + a11 * b21 + a21 * b22 + a31 * b23;
  dst[10] = a02 * b20 + a12 * b21 + a22 * b22 + a32 * b23;
  dst[11] = a03 * b20 + a13 * b21 + a23 * b22 + a33 * b23;
  dst[12] = a00 * b30 + a10 * b31 + a20 * b32 + a30 * b33;
  dst[13] = a01 * b30 + a11 * b31 + a21 * b32 + a31 * b33;
  dst[14] = a02 * b30 + a12 * b31 + a22 * b32 + a32 * b33;
  dst[15] = a03 * b30 + a13 * b31 + a23 * b32 + a33 * b33;
  return dst;
}

/** m * translation(v) */
function translate(m, v, dst) {
  dst = dst || mat4();
  var v0 = v[0], v1 = v[1], v2 = v[2];
  var m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3], m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7],
    m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11], m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
  if (m !== dst) for (var i = 0; i < 12; i++) dst[i] = m[i];
  dst[12] = m00 * v0 + m10 * v1 + m20 * v2 + m30;
  dst[13] = m01 * v0 + m11 * v1 + m21 * v2 + m31;
  dst[14] = m02 * v0 + m12 * v1 + m22 * v2 + m32;
  dst[15] = m03 * v0 + m13 * v1 + m23 * v2 + m33;
  return dst;
}

/** m * scaling(v) */
function scale(m, v, dst) {
  dst = dst || mat4();
  var v0 = v[0], v1 = v[1], v2 = v[2], i;
  for (i = 0; i < 4; i++) { dst[i] = v0 * m[i]; dst[4 + i] = v1 * m[4 + i]; dst[8 + i] = v2 * m[8 + i]; }
  if (m !== dst) for (i = 12; i < 16; i++) dst[i] = m[i];
  return dst;
}

function fromQuat(q) {
  var dst = mat4();
  var x = q[0], y = q[1], z = q[2], w = q[3];
  var x2 = x + x, y2 = y + y, z2 = z + z;
  var xx = x * x2, yx = y * x2, yy = y * y2, zx = z * x2, zy = z * y2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
  dst[0] = 1 - yy - zz; dst[1] = yx + wz; dst[2] = zx - wy; dst[3] = 0;
  dst[4] = yx - wz; dst[5] = 1 - xx - zz; dst[6] = zy + wx; dst[7] = 0;
  dst[8] = zx + wy; dst[9] = zy - wx; dst[10] = 1 - xx - yy; dst[11] = 0;
  dst[12] = 0; dst[13] = 0; dst[14] = 0; dst[15] = 1;
  return dst;
}

function transpose(m) {
  var d = mat4();
  for (var c = 0; c < 4; c++) for (var r = 0; r < 4; r++) d[c * 4 + r] = m[r * 4 + c];
  return d;
}

function inverse(m) {
  var dst = mat4();
  var m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3], m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7],
    m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11], m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
  var tmp0 = m22 * m33, tmp1 = m32 * m23, tmp2 = m12 * m33, tmp3 = m32 * m13, tmp4 = m12 * m23, tmp5 = m22 * m13,
    tmp6 = m02 * m33, tmp7 = m32 * m03, tmp8 = m02 * m23, tmp9 = m22 * m03, tmp10 = m02 * m13, tmp11 = m12 * m03,
    tmp12 = m20 * m31, tmp13 = m30 * m21, tmp14 = m10 * m31, tmp15 = m30 * m11, tmp16 = m10 * m21, tmp17 = m20 * m11,
    tmp18 = m00 * m31, tmp19 = m30 * m01, tmp20 = m00 * m21, tmp21 = m20 * m01, tmp22 = m00 * m11, tmp23 = m10 * m01;
  var t0 = (tmp0 * m11 + tmp3 * m21 + tmp4 * m31) - (tmp1 * m11 + tmp2 * m21 + tmp5 * m31);
  var t1 = (tmp1 * m01 + tmp6 * m21 + tmp9 * m31) - (tmp0 * m01 + tmp7 * m21 + tmp8 * m31);
  var t2 = (tmp2 * m01 + tmp7 * m11 + tmp10 * m31) - (tmp3 * m01 + tmp6 * m11 + tmp11 * m31);
  var t3 = (tmp5 * m01 + tmp8 * m11 + tmp11 * m21) - (tmp4 * m01 + tmp9 * m11 + tmp10 * m21);
  var d = 1 / (m00 * t0 + m10 * t1 + m20 * t2 + m30 * t3);
  dst[0] = d * t0; dst[1] = d * t1; dst[2] = d * t2; dst[3] = d * t3;
  dst[4] = d * ((tmp1 * m10 + tmp2 * m20 + tmp5 * m30) - (tmp0 * m10 + tmp3 * m20 + tmp4 * m30));
  dst[5] = d * ((tmp0 * m00 + tmp7 * m20 + tmp8 * m30) - (tmp1 * m00 + tmp6 * m20 + tmp9 * m30));
  dst[6] = d * ((tmp3 * m00 + tmp6 * m10 + tmp11 * m30) - (tmp2 * m00 + tmp7 * m10 + tmp10 * m30));
  dst[7] = d * ((tmp4 * m00 + tmp9 * m10 + tmp10 * m20) - (tmp5 * m00 + tmp8 * m10 + tmp11 * m20));
  dst[8] = d * ((tmp12 * m13 + tmp15 * m23 + tmp16 * m33) - (tmp13 * m13 + tmp14 * m23 + tmp17 * m33));
  dst[9] = d * ((tmp13 * m03 + tmp18 * m23 + tmp21 * m33) - (tmp12 * m03 + tmp19 * m23 + tmp20 * m33));
  dst[10] = d * ((tmp14 * m03 + tmp19 * m13 + tmp22 * m33) - (tmp15 * m03 + tmp18 * m13 + tmp23 * m33));
  dst[11] = d * ((tmp17 * m03 + tmp20 * m13 + tmp23 * m23) - (tmp16 * m03 + tmp21 * m13 + tmp22 * m23));
  dst[12] = d * ((tmp14 * m22 + tmp17 * m32 + tmp13 * m12) - (tmp16 * m32 + tmp12 * m12 + tmp15 * m22));
  dst[13] = d * ((tmp20 * m32 + tmp12 * m02 + tmp19 * m22) - (tmp18 * m22 + tmp21 * m32 + tmp13 * m02));
  dst[14] = d * ((tmp18 * m12 + tmp23 * m32 + tmp15 * m02) - (tmp22 * m32 + tmp14 * m02 + tmp19 * m12));
  dst[15] = d * ((tmp22 * m22 + tmp16 * m02 + tmp21 * m12) - (tmp20 * m12 + tmp23 * m22 + tmp17 * m02));
  return dst;
}

/** point transform with perspective divide */
function transformMat4(v, m) {
  var x = v[0], y = v[1], z = v[2];
  var w = (m[3] * x + m[7] * y + m[11] * z + m[15]) || 1;
  return vec3((m[0] * x + m[4] * y + m[8] * z + m[12]) / w, (m[1] * x + m[5] * y + m[9] * z + m[13]) / w,
    (m[2] * x + m[6] * y + m[10] * z + m[14]) / w);
}
function transformMat4Upper3x3(v, m) {
  var x = v[0], y = v[1], z = v[2];
  return vec3(x * m[0] + y * m[4] + z * m[8], x * m[1] + y * m[5] + z * m[9], x * m[2] + y * m[6] + z * m[10]);
}
function normalize(v) {
  var len = Math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  return len > 0.00001 ? vec3(v[0] / len, v[1] / len, v[2] / len) : vec3(0, 0, 0);
}
function length(v) { return Math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/** rotation part of a matrix -> quaternion (x, y, z, w) */
function quatFromMat(m) {
  var q = new Float32Array(4);
  var trace = m[0] + m[5] + m[10];
  if (trace > 0.0) {
    var root = Math.sqrt(trace + 1.0);
    q[3] = 0.5 * root;
    var invRoot = 0.5 / root;
    q[0] = (m[6] - m[9]) * invRoot; q[1] = (m[8] - m[2]) * invRoot; q[2] = (m[1] - m[4]) * invRoot;
  } else {
    var i = 0;
    if (m[5] > m[0]) i = 1;
    if (m[10] > m[i * 4 + i]) i = 2;
    var j = (i + 1) % 3, k = (i + 2) % 3;
    var r = Math.sqrt(m[i * 4 + i] - m[j * 4 + j] - m[k * 4 + k] + 1.0);
    q[i] = 0.5 * r;
    var ir = 0.5 / r;
    q[3] = (m[j * 4 + k] - m[k * 4 + j]) * ir;
    q[j] = (m[j * 4 + i] + m[i * 4 + j]) * ir;
    q[k] = (m[k * 4 + i] + m[i * 4 + k]) * ir;
  }
  return q;
}
function transformQuat(v, q) {
  var qx = q[0], qy = q[1], qz = q[2], w2 = q[3] * 2, x = v[0], y = v[1], z = v[2];
  var uvX = qy * z - qz * y, uvY = qz * x - qx * z, uvZ = qx * y - qy * x;
  return vec3(x + uvX * w2 + (qy * uvZ - qz * uvY) * 2, y + uvY * w2 + (qz * uvX - qx * uvZ) * 2,
    z + uvZ * w2 + (qx * uvY - qy * uvX) * 2);
}

module.exports = {
  vec3: vec3, identity: identity, create16: create16, clone: clone, mul: mul, translate: translate, scale: scale,
  fromQuat: fromQuat, transpose: transpose, inverse: inverse, transformMat4: transformMat4,
  transformMat4Upper3x3: transformMat4Upper3x3, normalize: normalize, length: length, quatFromMat: quatFromMat,
  transformQuat: transformQuat,
};
