'use strict';
/**
 * controller.js — the reference's camera controller (src/renderer/controller.ts) without the DOM: the same input
 * state and the same per-frame update, fed by any event source with on(name, handler) / off(name, handler)
 * (a Node EventEmitter, a websocket bridge from a browser, a test script). Events and their payloads keep the
 * DOM names the reference listens to:
 *   'keydown' / 'keyup'  { key }                        controller.ts:19-25
 *   'mousemove'          { movementX, movementY }       controller.ts:42-49 (pointer-lock deltas)
 *   'touchstart' / 'touchmove' / 'touchend'  { touches: [{clientX, clientY}, ...] }   controller.ts:51-113
 * update(deltaTime) is meant for renderer.addOnUpdate, like renderer.ts:535-538 wires it.
 */
var MOVE_SPEED = 2.0;                 // controller.ts:3
var ROTATE_SPEED = Math.PI / 18;      // controller.ts:4

// key -> (forward, right, up) direction of moveCamera, controller.ts:146-168; listed in the reference's order
var KEY_MOVES = [['w', [1, 0, 0]], ['s', [-1, 0, 0]], ['a', [0, -1, 0]], ['d', [0, 1, 0]], [' ', [0, 0, 1]]];

function Controller(renderer, source) {
  var self = this;
  this.renderer = renderer;
  this.source = source || null;
  this.pressed = {};
  this.mouse = { x: 0, y: 0 };
  this.lastTouch = null;
  this.twoFingers = false;
  this.pinch = 0;
  this.handlers = {
    keydown: function (e) { self.pressed[e.key] = true; },
    keyup: function (e) { self.pressed[e.key] = false; },
    mousemove: function (e) { self.mouse.x += e.movementX; self.mouse.y += e.movementY; },
    touchstart: function (e) {
      var t = e.touches;
      if (t.length === 1) { self.lastTouch = { x: t[0].clientX, y: t[0].clientY }; self.twoFingers = false; }
      else if (t.length === 2) { self.twoFingers = true; self.pinch = Math.hypot(t[0].clientX - t[1].clientX, t[0].clientY - t[1].clientY); }
    },
    touchmove: function (e) {
      var t = e.touches;
      if (t.length === 1 && self.lastTouch && !self.twoFingers) {          // one finger drags like the mouse
        self.mouse.x += t[0].clientX - self.lastTouch.x; self.mouse.y += t[0].clientY - self.lastTouch.y;
        self.lastTouch = { x: t[0].clientX, y: t[0].clientY };
      } else if (t.length === 2) {                                         // pinch moves along the view direction, at once
        var dist = Math.hypot(t[0].clientX - t[1].clientX, t[0].clientY - t[1].clientY);
        self.renderer.moveCamera((dist - self.pinch) * 0.001, 0, 0);
        self.pinch = dist;
      }
    },
    touchend: function (e) { if (e.touches.length === 0) { self.lastTouch = null; self.twoFingers = false; } },
  };
  this.handlers.touchcancel = this.handlers.touchend;
  if (this.source) Object.keys(this.handlers).forEach(function (name) { self.source.on(name, self.handlers[name]); });
}

/** feed one event by hand (no event source needed) */
Controller.prototype.handle = function (name, event) {
  if (!this.handlers[name]) throw new Error('Controller: unknown event "' + name + '"');
  this.handlers[name](event);
};

/** controller.ts:144-181 — speeds are per second, so the step scales with deltaTime */
Controller.prototype.update = function (deltaTime) {
  var step = MOVE_SPEED * deltaTime, r = this.renderer, self = this;
  KEY_MOVES.forEach(function (km) { if (self.pressed[km[0]]) r.moveCamera(km[1][0] * step, km[1][1] * step, km[1][2] * step); });
  if (this.pressed.Shift || this.pressed.q) r.moveCamera(0, 0, -step);
  if (this.mouse.x !== 0 || this.mouse.y !== 0) {
    r.rotateCamera(this.mouse.x * -ROTATE_SPEED * deltaTime, this.mouse.y * -ROTATE_SPEED * deltaTime);
    this.mouse.x = 0; this.mouse.y = 0;
  }
};

Controller.prototype.destroy = function () {
  var self = this;
  if (this.source && this.source.off) Object.keys(this.handlers).forEach(function (name) { self.source.off(name, self.handlers[name]); });
  else if (this.source && this.source.removeListener) Object.keys(this.handlers).forEach(function (name) { self.source.removeListener(name, self.handlers[name]); });
  this.source = null;
};

module.exports = { Controller: Controller, MOVE_SPEED: MOVE_SPEED, ROTATE_SPEED: ROTATE_SPEED };
