/*
 * ptmi_layout.h — byte layouts at the compute-pass boundary.
 *
 * These are the WGSL storage/uniform layouts of the reference's bind group 0
 * (reference: src/shader/pt.wgsl:7-78 structs, :104-110 bindings; host packing
 * in src/renderer/renderer.ts:242-355 via webgpu-utils). A caller hands the
 * library exactly the bytes it would have written into the WebGPU buffers.
 *
 * WGSL rules used: vec3f has align 16 / size 12 (a following scalar packs into
 * the spare 4 bytes); vec2f align 8; a struct's size rounds up to its largest
 * member alignment.
 *
 * Plain C (C99) and C++ compatible; no HIP or torch types.
 */
#ifndef PTMI_LAYOUT_H
#define PTMI_LAYOUT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define PTMI_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define PTMI_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* pt.wgsl:7-12 */
typedef struct ptmi_atlas_rect {
    uint32_t x, y, w, h;
} ptmi_atlas_rect;

/* pt.wgsl:14-26 — 128 B */
typedef struct ptmi_material {
    float base_color[3];        /*   0 */
    float metallic;             /*  12 */
    float roughness;            /*  16 */
    float _pad0[3];             /*  20 */
    float emission[3];          /*  32 */
    float emissive_strength;    /*  44 */
    float ior;                  /*  48 */
    float transmission;         /*  52 */
    ptmi_atlas_rect albedo_map;   /*  56 */
    ptmi_atlas_rect normal_map;   /*  72 */
    ptmi_atlas_rect pbr_map;      /*  88 */
    ptmi_atlas_rect emissive_map; /* 104 */
    float _pad1[2];             /* 120 */
} ptmi_material;

/* pt.wgsl:28-39 — 128 B */
typedef struct ptmi_triangle {
    float v0[3]; float _p0;     /*   0 */
    float v1[3]; float _p1;     /*  16 */
    float v2[3]; float _p2;     /*  32 */
    float n0[3]; float _p3;     /*  48 */
    float n1[3]; float _p4;     /*  64 */
    float n2[3]; float _p5;     /*  80 */
    float uv0[2];               /*  96 */
    float uv1[2];               /* 104 */
    float uv2[2];               /* 112 */
    uint32_t material_index;    /* 120 */
    uint32_t _p6;               /* 124 */
} ptmi_triangle;

/* pt.wgsl:67-78 — 48 B. Leaf <=> triangle_count > 0 (pt.wgsl:271); leaves
 * carry left = right = 0xFFFFFFFF (bvh.ts:87-88 stores -1 through Uint32Array). */
typedef struct ptmi_bvh_node {
    float aabb_min[3]; float _p0;   /*  0 */
    float aabb_max[3]; float _p1;   /* 16 */
    uint32_t left;                  /* 32 */
    uint32_t right;                 /* 36 */
    uint32_t triangle_offset;       /* 40 */
    uint32_t triangle_count;        /* 44 */
} ptmi_bvh_node;

/* pt.wgsl:41-51 — 48 B */
enum { PTMI_LIGHT_EMISSIVE = 0, PTMI_LIGHT_DIRECTIONAL = 1, PTMI_LIGHT_POINT = 2 };
typedef struct ptmi_light {
    float position[3];          /*  0  position, or direction for directional */
    uint32_t light_type;        /* 12 */
    float color[3];             /* 16 */
    float intensity;            /* 28 */
    uint32_t triangle_index;    /* 32  emissive lights only */
    uint32_t _pad[3];           /* 36 */
} ptmi_light;

/* pt.wgsl:53-65 — 96 B uniform, rewritten every frame (renderer.ts:403-413) */
typedef struct ptmi_camera {
    float position[3]; float _p0;   /*  0 */
    float forward[3];  float _p1;   /* 16 */
    float right[3];    float _p2;   /* 32 */
    float up[3];                    /* 48 */
    float fov;                      /* 60 */
    float aspect;                   /* 64 */
    uint32_t width;                 /* 68 */
    uint32_t height;                /* 72 */
    uint32_t frame_index;           /* 76 */
    float aperture;                 /* 80 */
    float focus_distance;           /* 84 */
    uint32_t _p3[2];                /* 88 */
} ptmi_camera;

/* outputBuffer: array<vec3f> — 16 B per pixel, index y*W+x, row 0 = image
 * bottom (pt.wgsl:104, :753; renderer.ts:272-279). */
#define PTMI_OUTPUT_STRIDE 16u

PTMI_STATIC_ASSERT(sizeof(ptmi_material) == 128, "Material is 128 B");
PTMI_STATIC_ASSERT(offsetof(ptmi_material, metallic) == 12, "Material.metallic");
PTMI_STATIC_ASSERT(offsetof(ptmi_material, roughness) == 16, "Material.roughness");
PTMI_STATIC_ASSERT(offsetof(ptmi_material, emission) == 32, "Material.emission");
PTMI_STATIC_ASSERT(offsetof(ptmi_material, emissive_strength) == 44, "Material.emissiveStrength");
PTMI_STATIC_ASSERT(offsetof(ptmi_material, ior) == 48, "Material.ior");
PTMI_STATIC_ASSERT(offsetof(ptmi_material, transmission) == 52, "Material.transmission");
PTMI_STATIC_ASSERT(offsetof(ptmi_material, albedo_map) == 56, "Material.albedoMap");
PTMI_STATIC_ASSERT(offsetof(ptmi_material, normal_map) == 72, "Material.normalMap");
PTMI_STATIC_ASSERT(offsetof(ptmi_material, pbr_map) == 88, "Material.pbrMap");
PTMI_STATIC_ASSERT(offsetof(ptmi_material, emissive_map) == 104, "Material.emissiveMap");

PTMI_STATIC_ASSERT(sizeof(ptmi_triangle) == 128, "Triangle is 128 B");
PTMI_STATIC_ASSERT(offsetof(ptmi_triangle, v1) == 16, "Triangle.v1");
PTMI_STATIC_ASSERT(offsetof(ptmi_triangle, v2) == 32, "Triangle.v2");
PTMI_STATIC_ASSERT(offsetof(ptmi_triangle, n0) == 48, "Triangle.n0");
PTMI_STATIC_ASSERT(offsetof(ptmi_triangle, n1) == 64, "Triangle.n1");
PTMI_STATIC_ASSERT(offsetof(ptmi_triangle, n2) == 80, "Triangle.n2");
PTMI_STATIC_ASSERT(offsetof(ptmi_triangle, uv0) == 96, "Triangle.uv0");
PTMI_STATIC_ASSERT(offsetof(ptmi_triangle, uv1) == 104, "Triangle.uv1");
PTMI_STATIC_ASSERT(offsetof(ptmi_triangle, uv2) == 112, "Triangle.uv2");
PTMI_STATIC_ASSERT(offsetof(ptmi_triangle, material_index) == 120, "Triangle.materialIndex");

PTMI_STATIC_ASSERT(sizeof(ptmi_bvh_node) == 48, "BVHNode is 48 B");
PTMI_STATIC_ASSERT(offsetof(ptmi_bvh_node, aabb_max) == 16, "BVHNode.aabb.max");
PTMI_STATIC_ASSERT(offsetof(ptmi_bvh_node, left) == 32, "BVHNode.left");
PTMI_STATIC_ASSERT(offsetof(ptmi_bvh_node, right) == 36, "BVHNode.right");
PTMI_STATIC_ASSERT(offsetof(ptmi_bvh_node, triangle_offset) == 40, "BVHNode.triangleOffset");
PTMI_STATIC_ASSERT(offsetof(ptmi_bvh_node, triangle_count) == 44, "BVHNode.triangleCount");

PTMI_STATIC_ASSERT(sizeof(ptmi_light) == 48, "Light is 48 B");
PTMI_STATIC_ASSERT(offsetof(ptmi_light, light_type) == 12, "Light.lightType");
PTMI_STATIC_ASSERT(offsetof(ptmi_light, color) == 16, "Light.color");
PTMI_STATIC_ASSERT(offsetof(ptmi_light, intensity) == 28, "Light.intensity");
PTMI_STATIC_ASSERT(offsetof(ptmi_light, triangle_index) == 32, "Light.triangleIndex");

PTMI_STATIC_ASSERT(sizeof(ptmi_camera) == 96, "Camera uniform is 96 B");
PTMI_STATIC_ASSERT(offsetof(ptmi_camera, forward) == 16, "Camera.forward");
PTMI_STATIC_ASSERT(offsetof(ptmi_camera, right) == 32, "Camera.right");
PTMI_STATIC_ASSERT(offsetof(ptmi_camera, up) == 48, "Camera.up");
PTMI_STATIC_ASSERT(offsetof(ptmi_camera, fov) == 60, "Camera.fov");
PTMI_STATIC_ASSERT(offsetof(ptmi_camera, aspect) == 64, "Camera.aspect");
PTMI_STATIC_ASSERT(offsetof(ptmi_camera, width) == 68, "Camera.width");
PTMI_STATIC_ASSERT(offsetof(ptmi_camera, height) == 72, "Camera.height");
PTMI_STATIC_ASSERT(offsetof(ptmi_camera, frame_index) == 76, "Camera.frameIndex");
PTMI_STATIC_ASSERT(offsetof(ptmi_camera, aperture) == 80, "Camera.aperture");
PTMI_STATIC_ASSERT(offsetof(ptmi_camera, focus_distance) == 84, "Camera.focusDistance");

#endif /* PTMI_LAYOUT_H */
