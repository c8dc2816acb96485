// Types of the Node host. The CPU-side scene types mirror the reference's
// src/renderer/gpu.ts:10-65 and src/renderer/bvh.ts:6-12.
export type Vec2 = ArrayLike<number>;
export type Vec3 = ArrayLike<number>;
export interface AtlasTexture { x: number; y: number; w: number; h: number; }
export interface MaterialCPU {
  baseColor: Vec3; metallic: number; roughness: number; emission: Vec3; emissiveStrength: number;
  ior: number; transmission: number;
  albedoMap: AtlasTexture; normalMap: AtlasTexture; pbrMap: AtlasTexture; emissiveMap: AtlasTexture;
}
export interface TriangleCPU {
  v0: Vec3; v1: Vec3; v2: Vec3; n0: Vec3; n1: Vec3; n2: Vec3; uv0: Vec2; uv1: Vec2; uv2: Vec2; materialIndex: number;
}
export interface CameraCPU {
  position: number[]; forward: number[]; right: number[]; up: number[]; fov: number; aspect: number;
  width: number; height: number; frameIndex: number; aperture: number; focusDistance: number;
}
export interface LightCPU { position: Vec3; lightType: number; color: Vec3; intensity: number; triangleIndex: number; }
export interface BVHNode { aabb: { min: Vec3; max: Vec3 }; left: number; right: number; triangleOffset: number; triangleCount: number; }
export interface SceneData { triangles: TriangleCPU[]; materials: MaterialCPU[]; bvhNodes: BVHNode[]; lights: LightCPU[]; }
export interface SceneBlobs { triangles: ArrayBuffer; materials: ArrayBuffer; bvhNodes: ArrayBuffer; lights: ArrayBuffer; }
export interface Atlas { data: ArrayBuffer; width: number; height: number; format: 1 | 2; }
export interface TraceOptions {
  maxBounces?: number; doMis?: number; tileY0?: number; tileY1?: number; framesPerBatch?: number;
  traversal?: 0 | 1 | 2 | 3; cull?: number; timing?: number; keepReferenceTree?: number;
  tileParts?: number; tilePart?: number; tileStrip?: number;
  /** include/ptmi.h: 1 = fast reciprocal / sqrt in `shade` (statistically, not bitwise, the same image); never the default */
  perfMode?: 0 | 1;
  /** 0 / 1 / 2 (library default): run the shadow kernel on a second stream beside the next bounce */
  overlap?: 0 | 1 | 2;
  /** read at loadModel: 0 (library default) / 1 the host builds the traversal hierarchy (SAH) / 2 the GPU does (linear BVH) */
  treeBuilder?: 0 | 1 | 2;
  /** read at loadModel: 0 (library default = 2) / 1 triangles are tested in the uploaded BVH's own leaves / 2 in the library's own
   *  leaves (a SAH hierarchy over the triangles; the winner is verified against its reference leaf, results unchanged) */
  leaves?: 0 | 1 | 2;
  /** leaves = 2: most triangles per own leaf (0 = library default) */
  leafTris?: number;
}
export interface Stats {
  paths: number; segments: number; shadowRays: number; frames: number; dispatches: number;
  gpuMs: number; extendMs: number; shadeMs: number; shadowMs: number; bvhDepth: number; traversalUsed: number;
  shadowTraced: number; uploadMs: number; framesPerBatchUsed: number; leavesUsed: number; leafTrisUsed: number;
  extendVariant: number; shadowVariant: number; verifyFailed: number;
}
export class Renderer {
  /** `devices: [0, 1, ...]` renders on several GPUs of one node behind one Renderer (include/ptmi.h ptmi_multi_*: rows dealt out as
   *  interleaved strips, one RCCL gather assembles the frame when it is read, or every `gatherEvery` frames). STATUS: checked with one
   *  device through RCCL and with several contexts on one device (`loopback: true`); more than one device over RCCL has never run —
   *  no machine this was built on has two GPUs. */
  constructor(options?: { device?: number; devices?: number[]; loopback?: boolean; gatherEvery?: number; maxFramesPerTick?: number;
                          width?: number; height?: number; options?: TraceOptions });
  camera: CameraCPU;
  addOnUpdate(callback: (deltaTime: number) => void): void;
  loadModel(model: string | SceneData | { blobs: SceneBlobs; atlas?: Atlas | null }, atlas?: Atlas): Promise<void>;
  renderFrame(frames?: number): void;
  start(): void;
  stop(): void;
  destroy(): void;
  resize(width: number, height: number): void;
  moveCamera(forward: number, right: number, up: number): void;
  rotateCamera(yaw: number, pitch: number): void;
  readOutput(): Float32Array;
  /** blit pass (blit.wgsl): tone-mapped RGBA8 canvas, row 0 = top */
  blit(): Uint8Array;
  setOptions(o: TraceOptions): void;
  getStats(): Stats;
}
export function setupRenderer(options?: { device?: number; width?: number; height?: number; model?: string; autoStart?: boolean; options?: TraceOptions; input?: InputSource }): Promise<Renderer>;
export const pack: {
  packTriangles(t: TriangleCPU[]): ArrayBuffer; packMaterials(m: MaterialCPU[]): ArrayBuffer;
  packBVH(n: BVHNode[]): ArrayBuffer; packLights(l: LightCPU[]): ArrayBuffer;
  packCamera(c: CameraCPU, out?: ArrayBuffer): ArrayBuffer; packScene(s: SceneData): SceneBlobs;
};
export function readSceneFile(path: string): { blobs: SceneBlobs; atlas: Atlas | null };
/** atlas.js — src/renderer/atlas.ts (PackedAtlas): the canvas as RGBA8 and as rgba16float texels, rects per material */
export interface PackedAtlas {
  texture: { width: number; height: number; rgba8: Uint8Array; data: Uint16Array; format: 1 };
  materials: Map<object, { albedoMap: AtlasTexture; normalMap: AtlasTexture; pbrMap: AtlasTexture; emissiveMap: AtlasTexture }>;
}
export const atlas: {
  potpack(boxes: { w: number; h: number; x?: number; y?: number }[]): { w: number; h: number; fill: number };
  packing(gltf: { materials: object[] }): PackedAtlas;
};
export function decodePNG(data: Uint8Array): { width: number; height: number; data: Uint8Array };
/** jpeg_decode.js — sequential and progressive Huffman JPEG, bit-identical to libjpeg-turbo's default decode */
export function decodeJPEG(data: Uint8Array): { width: number; height: number; data: Uint8Array };
/** controller.js — src/renderer/controller.ts without the DOM; events keep the DOM names and payload fields */
export interface InputSource { on(name: string, handler: (event: any) => void): void; off?(name: string, handler: (event: any) => void): void; }
export class Controller {
  constructor(renderer: { moveCamera(f: number, r: number, u: number): void; rotateCamera(yaw: number, pitch: number): void }, source?: InputSource);
  handle(name: 'keydown' | 'keyup' | 'mousemove' | 'touchstart' | 'touchmove' | 'touchend' | 'touchcancel', event: any): void;
  update(deltaTime: number): void;
  destroy(): void;
}
