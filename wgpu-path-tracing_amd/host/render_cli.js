#!/usr/bin/env node
'use strict';
/**
 * render_cli.js — drives the Renderer the way the reference's frame loop does
 * (renderer.ts:415-454: one dispatch per frame, frameIndex++), headless:
 *   node render_cli.js <scene.ptscene> <out.f32> [--width W --height H --frames N --bounces B --mis 0|1
 *                       --aperture A --focus F --batch K --png out.png]
 * --batch K traces K frames per dispatch instead of one. Writes W*H*4 float32 (the output buffer)
 * and prints one JSON line with the statistics.
 */
var fs = require('fs');
var host = require('./renderer');

function arg(name, dflt) {
  var i = process.argv.indexOf('--' + name);
  return i >= 0 ? Number(process.argv[i + 1]) : dflt;
}

var scenePath = process.argv[2], outPath = process.argv[3];
if (!scenePath || !outPath) { console.error('usage: render_cli.js scene.ptscene out.f32 [options]'); process.exit(2); }
var W = arg('width', 256), H = arg('height', 256), frames = arg('frames', 16), batch = arg('batch', 1);

var r = new host.Renderer({ width: W, height: H, options: { maxBounces: arg('bounces', 8), doMis: arg('mis', 1) } });
r.camera.aperture = arg('aperture', r.camera.aperture);
r.camera.focusDistance = arg('focus', r.camera.focusDistance);
r.loadModel(scenePath).then(function () {
  var t0 = Date.now();
  while (r.frameIndex < frames) r.renderFrame(Math.min(batch, frames - r.frameIndex));
  var out = r.readOutput();
  var ms = Date.now() - t0;
  fs.writeFileSync(outPath, Buffer.from(out.buffer));
  var pi = process.argv.indexOf('--png');
  if (pi >= 0) fs.writeFileSync(process.argv[pi + 1], require('./png').encodePNG(r.blit(), W, H));
  var st = r.getStats();
  st.wallMs = ms; st.width = W; st.height = H; st.frames = frames;
  console.log(JSON.stringify(st));
  r.destroy();
}).catch(function (e) { console.error(String(e && e.stack || e)); process.exit(1); });
