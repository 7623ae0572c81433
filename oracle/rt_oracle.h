/* rt_oracle.h — entry points of the CPU oracle (test infrastructure only; see
 * rt_oracle.cpp). Same POD types as include/rt2022.h. */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include "../include/rt2022.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rto_hit_record {       /* HitRecord, hittable/mod.rs:18-26 */
    int32_t  hit;
    int32_t  front_face;
    double   p[3];
    double   normal[3];
    double   t, u, v;
    uint32_t mat;
    uint32_t rng_draws;
} rto_hit_record;

enum rto_math_op { RTO_SIN = 0, RTO_COS = 1, RTO_ACOS = 2, RTO_ATAN2 = 3, RTO_LOG = 4, RTO_SQRT = 5, RTO_DIV = 6 };

/* CPU twin of rt_render (main.rs:109-162 threading scheme with n_threads workers). */
int rt_render_cpu(const rt_scene_desc *scene, const rt_camera *cam, const rt_params *params,
                  double *out_rgb_sum, rt_stats *stats, int n_threads);
void rto_write_color(const double rgb_sum[3], int32_t spp, uint8_t out[3]);
const char *rto_last_error(void);

int rto_hit(const rt_scene_desc *scene, uint32_t ref, const double ray[7], double t_min, double t_max,
            uint64_t rng_state, rto_hit_record *out, rt_stats *stats);
int rto_ray_color(const rt_scene_desc *scene, const double ray[7], const double background[3], double t_min,
                  int depth, uint64_t rng_state, double out_rgb[3], rt_stats *stats);
int rto_get_ray(const rt_camera *cam, double s, double t, uint64_t rng_state, double out_ray[7]);
int rto_texture_value(const rt_scene_desc *scene, uint32_t tex, double u, double v, const double p[3], double out_rgb[3]);
double rto_perlin_noise(const rt_perlin *pl, const double p[3]);
double rto_perlin_turb(const rt_perlin *pl, const double p[3], int depth);
double rto_lights_pdf_value(const rt_scene_desc *scene, const double o[3], const double v[3]);
int rto_lights_random(const rt_scene_desc *scene, const double o[3], uint64_t rng_state, double out_dir[3]);
int rto_scatter(const rt_scene_desc *scene, uint32_t mat, const double ray_in[7], const rto_hit_record *rec_in,
                uint64_t rng_state, double out_ray[7], double out_attenuation[3], double out_emitted[3]);
double rto_math(int op, double a, double b);
/* 1 if this library was built with -DRTO_LIBM (`make libm`: the render path calls the platform libm). */
int rto_uses_libm(void);
/* 1 if this library was built with -DRTO_OWN_MATH (`make own`: oracle/rto_math.h instead of the product's rt_math.h). */
int rto_uses_own_math(void);
/* The transcendental as the render path of THIS build calls it (rt_math.h, or the platform libm under -DRTO_LIBM). */
double rto_path_math(int op, double a, double b);
void rto_math_array(int op, const double *a, const double *b, double *out, uint64_t n);
void rto_rng_u64(uint64_t state, uint64_t *out, uint64_t n);
void rto_rng_f64(uint64_t state, double *out, uint64_t n);
void rto_rng_range(uint64_t state, double lo, double hi, double *out, uint64_t n);
void rto_rng_index(uint64_t state, uint64_t bound, uint64_t *out, uint64_t n);
uint64_t rto_path_key(uint64_t seed, uint32_t frame, uint64_t pixel, uint32_t sample);

#ifdef __cplusplus
}
#endif
#endif
