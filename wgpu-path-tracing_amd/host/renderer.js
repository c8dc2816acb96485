'use strict';
/**
 * renderer.js — Node host with the reference Renderer's public surface
 * (src/renderer/renderer.ts:18-511: loadModel, start, stop, destroy, resize, moveCamera,
 * rotateCamera, addOnUpdate, camera) whose compute pass runs on the MI355X through the
 * N-API addon instead of WebGPU:
 *
 *   createBuffers + createBindGroups (renderer.ts:242-355, :368-381) -> uploadScene / uploadAtlas / resize
 *   updateCamera                      (renderer.ts:403-413)          -> packCamera (96-byte uniform)
 *   compute pass dispatch             (renderer.ts:421-431)          -> addon.dispatch(ctx, camera, frames)
 *
 * options.devices = [ordinal, ...] puts several GPUs of the node behind the same Renderer (include/ptmi.h ptmi_multi_*): the
 * frame's rows are dealt out as interleaved strips, every device accumulates its own, and the frame is assembled on the first
 * device by one RCCL gather — when it is read (readOutput / blit), and every options.gatherEvery frames for a preview.
 *
 * The frame loop (start) paces itself: the reference's is paced by requestAnimationFrame (renderer.ts:456-473); here a tick
 * first waits until at most one earlier dispatch is unfinished (two in flight with the new one), and while the camera stands
 * still it doubles the frames per dispatch up to options.maxFramesPerTick (64) — ptmi_dispatch(camera, n) IS n single-frame
 * dispatches (include/ptmi.h), so the image is the reference's, at the rate of large batches; any camera change drops to 1.
 *
 * The blit pass, tweakpane stats and the DOM controller are out of scope (SURVEY.md §8).
 * Plain JavaScript (Node >= 12: no optional chaining), typed by index.d.ts.
 */
var path = require('path');
var pack = require('./pack');
var sceneFile = require('./scene_file');

var addon = null;
function loadAddon() {
  if (!addon) addon = require(path.join(__dirname, 'addon', 'ptmi_napi.node'));   // throws if not built
  return addon;
}

var MAX_FRAMES = -1;                       // renderer.ts:16

// the addon's single-device and multi-device entry points under one set of names
function deviceApi(addon, devices, loopback) {
  if (!devices) return { multi: false, create: function (d) { return addon.create(d); }, destroy: addon.destroy,
    uploadScene: addon.uploadScene, uploadAtlas: addon.uploadAtlas, resize: addon.resize, setOptions: addon.setOptions,
    dispatch: addon.dispatch, synchronize: addon.synchronize, throttle: addon.throttle, readOutput: addon.readOutput,
    writeOutput: addon.writeOutput, blit: addon.blit, getStats: addon.getStats, gather: function () {} };
  return { multi: true, create: function () { return addon.multiCreate(devices, loopback ? 1 : 0); }, destroy: addon.multiDestroy,
    uploadScene: addon.multiUploadScene, uploadAtlas: addon.multiUploadAtlas, resize: addon.multiResize,
    setOptions: addon.multiSetOptions, dispatch: addon.multiDispatch, synchronize: addon.multiSynchronize,
    throttle: addon.multiThrottle, readOutput: addon.multiReadOutput, writeOutput: addon.multiWriteOutput, blit: addon.multiBlit,
    getStats: addon.multiGetStats, gather: addon.multiGather };
}

function Renderer(options) {
  options = options || {};
  this.addon = loadAddon();
  this.api = deviceApi(this.addon, options.devices, options.loopback);
  this.ctx = this.api.create(options.device || 0);     // throws without a gfx950 device: there is no CPU path
  this.gatherEvery = options.gatherEvery || 0;          // several devices: assemble the frame every so many frames (0: when it is read)
  this.sinceGather = 0;
  this.maxFramesPerTick = options.maxFramesPerTick || 64;
  this.framesPerTick = 1;
  this.width = options.width || 800;
  this.height = options.height || 600;
  this.frameIndex = 0;
  this.onUpdateTasks = [];
  this.timer = null;
  this.lastTime = 0;
  this.sceneLoaded = false;
  this.cameraBytes = new ArrayBuffer(pack.CAMERA_SIZE);
  this.setupCamera();
  this.api.resize(this.ctx, this.width, this.height);
  if (options.options) this.api.setOptions(this.ctx, options.options);
}

/** renderer.ts:136-150 */
Renderer.prototype.setupCamera = function () {
  this.camera = {
    position: [0, 1.0, 2.8], forward: [0, 0, -1], right: [1, 0, 0], up: [0, 1, 0],
    fov: Math.PI / 3, aspect: this.width / this.height, width: this.width, height: this.height,
    frameIndex: 0, focusDistance: 5.0, aperture: 0.001,
  };
};

Renderer.prototype.addOnUpdate = function (callback) { this.onUpdateTasks.push(callback); };

/**
 * renderer.ts:130-134. `model` is a .glb path (loaded and prepared like loader.ts + gpu.ts do), a
 * .ptscene path, {blobs, atlas} from readSceneFile / prepareScene, or a SceneData object
 * (gpu.ts:60-65) which is packed here like renderer.ts:282-320 does.
 */
Renderer.prototype.loadModel = function (model, atlas) {
  var self = this;
  return new Promise(function (resolve) {
    var blobs;
    if (typeof model === 'string' && /\.glb$/i.test(model)) {
      var prepared = require('./scene_prep').prepareScene(require('./gltf').loadGLB(model));
      blobs = prepared.blobs; atlas = atlas || prepared.atlas; self.sceneInfo = prepared;
    } else if (typeof model === 'string') {
      var f = sceneFile.readSceneFile(model);
      blobs = f.blobs; atlas = atlas || f.atlas;
    } else if (model.blobs) {
      blobs = model.blobs; atlas = atlas || model.atlas;
    } else {
      blobs = pack.packScene(model);
    }
    self.api.uploadScene(self.ctx, blobs.triangles, blobs.materials, blobs.bvhNodes, blobs.lights);
    if (atlas) self.api.uploadAtlas(self.ctx, atlas.data, atlas.width, atlas.height, atlas.format || 1);
    else self.api.uploadAtlas(self.ctx, null, 0, 0, 0);
    self.sceneLoaded = true;
    self.resetOutputBuffer(false);
    resolve();
  });
};

/** renderer.ts:357-366 */
Renderer.prototype.resetOutputBuffer = function (restart) {
  this.frameIndex = 0;
  this.camera.frameIndex = 0;
  this.framesPerTick = 1;                   // the picture changed: back to one frame per tick, for the shortest latency
  if (restart !== false && this.timer === null && this.sceneLoaded) this.start();
};

/** renderer.ts:403-413 */
Renderer.prototype.updateCamera = function () {
  this.camera.frameIndex = this.frameIndex;
  pack.packCamera(this.camera, this.cameraBytes);
};

/** renderer.ts:415-454 (compute pass only). frames > 1 traces that many consecutive frames in one call. */
Renderer.prototype.renderFrame = function (frames) {
  frames = frames || 1;
  this.updateCamera();
  this.api.dispatch(this.ctx, this.cameraBytes, frames);
  this.frameIndex += frames;
  if (this.api.multi && this.gatherEvery > 0) {
    this.sinceGather += frames;
    if (this.sinceGather >= this.gatherEvery) { this.api.gather(this.ctx); this.sinceGather = 0; }
  }
};

/** renderer.ts:456-473 — requestAnimationFrame becomes setImmediate */
Renderer.prototype.start = function () {
  var self = this;
  this.lastTime = Date.now();
  var animate = function () {
    // Back-pressure without blocking the event loop (the reference's requestAnimationFrame loop never blocks: input events keep
    // flowing): POLL how many dispatches are unfinished and come back on the next turn of the loop while more than one is — at
    // most two in flight with the one a tick enqueues. A turn that only waits is not a tick: no update task runs.
    if (self.sceneLoaded && self.api.throttle(self.ctx, 0xFFFFFFFF) > 1) {
      self.throttledTurns = (self.throttledTurns || 0) + 1;
      if (self.timer !== null) self.timer = setImmediate(animate);
      return;
    }
    var now = Date.now();
    var dt = (now - self.lastTime) / 1000;
    self.lastTime = now;
    for (var i = 0; i < self.onUpdateTasks.length; i++) self.onUpdateTasks[i](dt);      // may move the camera: framesPerTick = 1
    if (self.timer === null) return;        // an update task stopped the loop
    if (MAX_FRAMES === -1 || self.frameIndex < MAX_FRAMES) {
      var n = self.framesPerTick;
      if (MAX_FRAMES !== -1) n = Math.min(n, MAX_FRAMES - self.frameIndex);
      self.renderFrame(n);
      self.framesPerTick = Math.min(self.framesPerTick * 2, self.maxFramesPerTick);     // still camera: larger batches
    }
    if (self.timer !== null) self.timer = setImmediate(animate);
  };
  this.timer = setImmediate(animate);
};

Renderer.prototype.stop = function () {
  if (this.timer !== null) { clearImmediate(this.timer); this.timer = null; }
};

/** renderer.ts:482-494 */
Renderer.prototype.destroy = function () {
  this.stop();
  if (this.ctx) { this.api.destroy(this.ctx); this.ctx = null; }
};

/** renderer.ts:496-510 */
Renderer.prototype.resize = function (width, height) {
  this.width = width; this.height = height;
  this.camera.aspect = width / height;
  this.camera.width = width; this.camera.height = height;
  this.frameIndex = 0;
  this.framesPerTick = 1;
  this.api.resize(this.ctx, width, height);
};

/** renderer.ts:152-170 */
Renderer.prototype.moveCamera = function (forward, right, up) {
  var c = this.camera;
  for (var k = 0; k < 3; k++) c.position[k] += right * c.right[k] + forward * c.forward[k] + up * c.up[k];
  this.resetOutputBuffer();
};

function normalize(v) { var l = Math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); return [v[0] / l, v[1] / l, v[2] / l]; }
function cross(a, b) { return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]; }

/** renderer.ts:172-201: yaw about +Y, then the clamped pitch delta about +X */
Renderer.prototype.rotateCamera = function (yaw, pitch) {
  var c = this.camera;
  var currentPitch = Math.asin(c.forward[1]);
  var lim = (Math.PI / 2) * 0.99;
  var newPitch = Math.max(Math.min(currentPitch + pitch, lim), -lim);
  var dp = newPitch - currentPitch;
  var f = c.forward;
  var cx = Math.cos(dp), sx = Math.sin(dp);                 // v1 = Rx(dp) v
  var v1 = [f[0], cx * f[1] - sx * f[2], sx * f[1] + cx * f[2]];
  var cy = Math.cos(yaw), sy = Math.sin(yaw);               // v2 = Ry(yaw) v1
  c.forward = normalize([cy * v1[0] + sy * v1[2], v1[1], -sy * v1[0] + cy * v1[2]]);
  c.right = normalize(cross(c.forward, [0, 1, 0]));
  c.up = normalize(cross(c.right, c.forward));
  this.resetOutputBuffer();
};

/** Output buffer (binding 0): width*height float4, row 0 = image bottom. Synchronises. */
Renderer.prototype.readOutput = function () {
  var out = new Float32Array(this.width * this.height * 4);
  this.api.readOutput(this.ctx, out);
  return out;
};

/** The reference's blit pass (renderer.ts:434-449, blit.wgsl): tone-mapped 8-bit canvas, row 0 = top. */
Renderer.prototype.blit = function () {
  var out = new Uint8Array(this.width * this.height * 4);
  this.api.blit(this.ctx, out);
  return out;
};

Renderer.prototype.setOptions = function (o) { this.api.setOptions(this.ctx, o); };
Renderer.prototype.getStats = function () { return this.api.getStats(this.ctx); };
/** several devices: assemble the frame on the first one now (readOutput / blit do it themselves) */
Renderer.prototype.gather = function () { this.api.gather(this.ctx); this.sinceGather = 0; };
Renderer.prototype.synchronize = function () { this.api.synchronize(this.ctx); };

/** renderer.ts:513-558 without the canvas: create, load, (optionally) start; options.input (an event source,
 *  see controller.js) gets a Controller whose update runs every frame, like renderer.ts:554-555 */
function setupRenderer(options) {
  var r = new Renderer(options);
  if (options && options.input) {
    var controller = new (require('./controller').Controller)(r, options.input);
    r.controller = controller;
    r.addOnUpdate(function (deltaTime) { controller.update(deltaTime); });
  }
  if (!options || !options.model) return Promise.resolve(r);
  return r.loadModel(options.model).then(function () { if (options.autoStart) r.start(); return r; });
}

module.exports = { Renderer: Renderer, setupRenderer: setupRenderer, pack: pack, readSceneFile: sceneFile.readSceneFile,
  atlas: require('./atlas'), decodePNG: require('./png_decode').decodePNG,
  decodeJPEG: require('./jpeg_decode').decodeJPEG, Controller: require('./controller').Controller };
