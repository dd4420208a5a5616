/* capi_demo.c -- the C ABI of libfleet_hip.so used from plain C, no Python and no PyTorch in the process.
 *
 *   gcc -I include examples/capi_demo.c -o capi_demo -L fleetrl_amd -l:libfleet_hip.so -Wl,-rpath,$PWD/fleetrl_amd -lm
 *   ./capi_demo <params.bin> <tables.bin> <steps>
 *
 * params.bin: one FleetParams struct as written by the caller (tests/test_capi_gpu.py writes the ctypes mirror).
 * tables.bin: the FleetTables columns back to back in declaration order (there u8[T*N], time_left f32[T*N],
 *             soc_on_return f64[T*N], delu/tariff/prc/trc/load/pv f64[T] each, hour/minute/month/weekday u8[T] each).
 * Drives `steps` steps with a fixed action pattern and prints the sum of rewards, the number of finished episodes and the
 * final SOC sum -- the same numbers the test computes through the Python front end.  With -DFLEET_DEMO_DEVICE_TAPE (see below) it
 * also replays a device-resident action tape through HIP launches and through the library's own queue and compares the results.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fleet_hip.h"
#ifdef FLEET_DEMO_DEVICE_TAPE
#include <hip/hip_runtime_api.h>
#endif

static void* slurp(const char* path, size_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  *len = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  void* buf = malloc(*len);
  if (fread(buf, 1, *len, f) != *len) { perror("read"); exit(2); }
  fclose(f);
  return buf;
}

int main(int argc, char** argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s params.bin tables.bin steps\n", argv[0]); return 2; }
  size_t plen, tlen;
  FleetParams* p = (FleetParams*)slurp(argv[1], &plen);
  unsigned char* blob = (unsigned char*)slurp(argv[2], &tlen);
  const int steps = atoi(argv[3]);
  if (plen != sizeof(FleetParams) || p->struct_bytes != (int)sizeof(FleetParams)) { fprintf(stderr, "FleetParams size mismatch\n"); return 2; }
  const size_t T = (size_t)p->table_rows, N = (size_t)p->num_cars, E = (size_t)p->num_envs, TN = T * N;

  FleetTables t;
  memset(&t, 0, sizeof t);
  unsigned char* q = blob;
  t.there = q;                          q += TN;
  t.time_left = (const float*)q;        q += TN * 4;
  t.soc_on_return = (const double*)q;   q += TN * 8;
  t.delu = (const double*)q;            q += T * 8;
  t.tariff = (const double*)q;          q += T * 8;
  t.prc = (const double*)q;             q += T * 8;
  t.trc = (const double*)q;             q += T * 8;
  t.load = (const double*)q;            q += T * 8;
  t.pv = (const double*)q;              q += T * 8;
  t.hour = q;                           q += T;
  t.minute = q;                         q += T;
  t.month = q;                          q += T;
  t.weekday = q;                        q += T;
  t.time_feat = NULL; /* the library computes the six sin/cos features itself */
  if ((size_t)(q - blob) != tlen) { fprintf(stderr, "tables.bin: %zu bytes, expected %zu\n", tlen, (size_t)(q - blob)); return 2; }

  fleet_handle h = NULL;
  if (fleet_create(p, &t, 0, &h) != FLEET_OK) { fprintf(stderr, "fleet_create: %s\n", fleet_last_error(NULL)); return 1; }
  const int D = fleet_obs_dim(p);
  float* obs = (float*)malloc(E * (size_t)D * sizeof(float));
  float* act = (float*)malloc(E * N * sizeof(float));
  double* rew = (double*)malloc(E * sizeof(double));
  unsigned char* done = (unsigned char*)malloc(E);
  if (fleet_reset_host(h, NULL, obs) != FLEET_OK) { fprintf(stderr, "reset: %s\n", fleet_last_error(h)); return 1; }

  double reward_sum = 0.0;
  long episodes = 0;
  for (int s = 0; s < steps; ++s) {
    for (size_t i = 0; i < E * N; ++i) act[i] = (float)(((int)((i * 7 + (size_t)s * 13) % 21) - 8) / 12.0); /* in [-0.67, 1] */
    if (fleet_step_host(h, act, FLEET_ACT_F32, obs, rew, done, NULL) != FLEET_OK) { fprintf(stderr, "step: %s\n", fleet_last_error(h)); return 1; }
    for (size_t e = 0; e < E; ++e) { reward_sum += rew[e]; episodes += done[e]; }
  }
  double* soc = (double*)malloc(E * N * sizeof(double));
  if (fleet_get(h, FLEET_F_SOC, soc) != FLEET_OK || fleet_check_errors(h) != FLEET_OK) { fprintf(stderr, "get: %s\n", fleet_last_error(h)); return 1; }
  double soc_sum = 0.0;
  for (size_t i = 0; i < E * N; ++i) soc_sum += soc[i];
  printf("%.17g %ld %.17g %d\n", reward_sum, episodes, soc_sum, D);
  fleet_destroy(h);

#ifdef FLEET_DEMO_DEVICE_TAPE
  /* Device-pointer entry points, still without PyTorch: two handles from the same parameters replay one recorded action tape of
   * `steps` steps -- one through HIP launches, one through the library's own AQL queue (FLEET_LAUNCH_DIRECT) -- and must end in the
   * same state, bit for bit.  (Needs the HIP runtime for the device buffers: build with -DFLEET_DEMO_DEVICE_TAPE
   * -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -lamdhip64.) */
  {
    const int L = 16;
    float* tape_h = (float*)malloc((size_t)L * E * N * sizeof(float));
    for (size_t i = 0; i < (size_t)L * E * N; ++i) tape_h[i] = (float)(((int)((i * 11) % 23) - 9) / 13.0);
    double* soc2[2];
    int queues = -1;
    for (int mode = 0; mode < 2; ++mode) {
      fleet_handle g = NULL;
      void *tape_d, *obs_d, *rew_d, *done_d;
      if (fleet_create(p, &t, 0, &g) != FLEET_OK) { fprintf(stderr, "fleet_create: %s\n", fleet_last_error(NULL)); return 1; }
      if (hipMalloc(&tape_d, (size_t)L * E * N * 4) != hipSuccess || hipMalloc(&obs_d, E * (size_t)D * 4) != hipSuccess ||
          hipMalloc(&rew_d, E * 8) != hipSuccess || hipMalloc(&done_d, E) != hipSuccess ||
          hipMemcpy(tape_d, tape_h, (size_t)L * E * N * 4, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "hipMalloc\n"); return 1; }
      if (fleet_reset_dev(g, NULL, (float*)obs_d) != FLEET_OK ||
          fleet_run_tape_dev(g, steps, tape_d, L, FLEET_ACT_F32, (float*)obs_d, (double*)rew_d, (unsigned char*)done_d,
                             mode ? FLEET_LAUNCH_DIRECT : FLEET_LAUNCH_EAGER) != FLEET_OK ||
          fleet_synchronize(g) != FLEET_OK) { fprintf(stderr, "tape run (%d): %s\n", mode, fleet_last_error(g)); return 1; }
      if (mode) queues = fleet_direct_queues(g);
      soc2[mode] = (double*)malloc(E * N * sizeof(double));
      if (fleet_get(g, FLEET_F_SOC, soc2[mode]) != FLEET_OK || fleet_check_errors(g) != FLEET_OK) { fprintf(stderr, "get: %s\n", fleet_last_error(g)); return 1; }
      fleet_destroy(g);
      hipFree(tape_d); hipFree(obs_d); hipFree(rew_d); hipFree(done_d);
    }
    printf("direct_queue_matches %d queues %d\n", memcmp(soc2[0], soc2[1], E * N * sizeof(double)) == 0, queues);
  }
#endif
  return 0;
}
