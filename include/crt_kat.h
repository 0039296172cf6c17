/* crt_kat.h — record layouts of the known-answer-test (KAT) entry points.
 *
 * Shared by `crt_hip_kat` (device functions, include/crt_hip.h) and by the CPU oracle's
 * `orc_kat` so a test feeds the SAME float records to both and compares the outputs.
 * All records are arrays of 32-bit floats; integer fields travel as raw bits.
 *
 * Each function id names the reference function(s) it exercises (reference tree paths):
 */
#ifndef CRT_KAT_H
#define CRT_KAT_H

enum {
    /* backends/embree/disney_bsdf.ih:311-359 disney_brdf + disney_pdf
     * in  (29): mat[14] n[3] w_o[3] w_i[3] v_x[3] v_y[3]
     * out  (4): brdf.rgb, pdf */
    CRT_KAT_DISNEY_EVAL = 1,
    /* disney_bsdf.ih:364-429 sample_disney_brdf
     * in  (27): mat[14] n[3] w_o[3] v_x[3] v_y[3] rng_state(bits)
     * out  (8): f.rgb, w_i[3], pdf, rng_state_after(bits) */
    CRT_KAT_DISNEY_SAMPLE = 2,
    /* backends/embree/lights.ih:26-69 sample_quad_light_position, quad_light_pdf,
     * quad_intersect
     * in  (28): light[20] orig[3] dir[3] samples[2]
     * out  (9): sampled_pos[3], pdf(light, sampled_pos, orig, dir), hit(0/1), t, hit_pos[3] */
    CRT_KAT_LIGHT = 3,
    /* backends/embree/texture2d.ih:39-83 texture + texture_channel (needs a scene)
     * in   (4): tex_id(bits) u v channel(bits)
     * out  (5): rgba[4], channel_value */
    CRT_KAT_TEXTURE = 4,
    /* backends/embree/render_embree.ispc:184-196 miss_shader
     * in   (3): dir[3]        out (3): rgb */
    CRT_KAT_MISS = 5,
    /* backends/embree/util.ih:32-46 ortho_basis
     * in   (3): n[3]          out (6): v_x[3] v_y[3] */
    CRT_KAT_ORTHO_BASIS = 6,
    /* linear -> sRGB8 as every non-ISPC backend does it
     * (backends/embree_sycl/render_embree_kernel.inl:312-315, util.ih:17-22)
     * in   (1): x             out (1): (float)uint8 */
    CRT_KAT_SRGB8 = 7,
    /* backends/embree/lcg_rng.ih:4-59 get_rng + lcg_random + lcg_randomf
     * in   (2): pixel_id(bits) frame_id(bits)
     * out (17): state0(bits), then 8 x { lcg_random(bits), the float lcg_randomf gives for it } */
    CRT_KAT_RNG = 8,
    /* render_embree.ispc:79-103 unpack_material (needs a scene)
     * in   (3): material_id(bits) u v      out (14): DisneyMaterial fields */
    CRT_KAT_UNPACK_MATERIAL = 9,
    /* render_embree.ispc:105-181 sample_direct_light around its two occlusion queries (needs a scene: its lights):
     * light pick, light sample, both pdfs, MIS weights, the BSDF-sample branch; both rays taken as unoccluded
     * in  (30): mat[14] n[3] w_o[3] v_x[3] v_y[3] hit_p[3] rng_state(bits)
     * out (17): c_a[3] (contribution of the light-sample ray, 0 if a pdf < EPSILON), light_dir[3], light_dist,
     *           has_b(0/1), c_b[3], w_i_b[3], light_dist_b (zeros without a second ray), rng_state_after(bits),
     *           occlusion rays counted (1 or 2) */
    CRT_KAT_NEE = 10,
    /* render_embree.ispc:327-335 Russian roulette, as the path loop applies it once bounce > 3: q = max(0.05, 1 - max(tp.x,
     * max(tp.y, tp.z))), one lcg_randomf draw, termination if it is < q, else tp /= 1 - q. max(a, b) is the reference's
     * `a < b ? b : a` (sycl::max as embree_sycl/render_embree_kernel.inl:284-287 uses it; NOT fmax: a NaN first operand stays)
     * in   (4): tp[3] rng_state(bits)
     * out  (6): terminated(0/1), tp_after[3] (tp itself if terminated), rng_state_after(bits), q */
    CRT_KAT_ROULETTE = 11
};

#define CRT_KAT_MAX_IN 32
#define CRT_KAT_MAX_OUT 20

#endif
