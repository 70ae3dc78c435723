/* selfplay_client.c -- a plain C consumer of libazulhip.so (no Python, no torch, no C++): what a cgo / JNI / FFI binding of
 * include/azul_hip.h amounts to.  Plays N games x T env moves of random-agent self-play and dumps, per move and game,
 * action / reward / done and the final 128-byte records as raw little-endian bytes to stdout's file argument; the test
 * (tests/test_c_abi_client.py) diffs that file against the oracle.
 *
 *   gcc -std=c11 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tests/c_client/selfplay_client.c \
 *       -L azul_deep_reinforcement_learning_amd -lazulhip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,... -o client
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "azul_hip.h"

#define CHECK_AZ(x) do { int rc_ = (x); if (rc_ != AZUL_SUCCESS) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, azul_last_error_string()); return 2; } } while (0)
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s n_games n_steps seed_base out_file\n", argv[0]); return 1; }
    const int n = atoi(argv[1]), t = atoi(argv[2]);
    const uint64_t seed_base = strtoull(argv[3], NULL, 10);
    azul_batch_t *b = NULL;
    CHECK_AZ(azul_batch_create(&b, n, AZUL_FIRST_RANDOM, AZUL_POOL_LID));
    CHECK_AZ(azul_batch_seed(b, seed_base, NULL, NULL));
    CHECK_AZ(azul_batch_runner_init(b, NULL, NULL, NULL));      /* GameRunner()  */
    CHECK_AZ(azul_batch_runner_init(b, NULL, NULL, NULL));      /* reset()       */
    int32_t *action_dev, *reward_dev;
    uint8_t *done_dev;
    const size_t cells = (size_t)n * (size_t)t;
    CHECK_HIP(hipMalloc((void **)&action_dev, cells * 4));
    CHECK_HIP(hipMalloc((void **)&reward_dev, cells * 4));
    CHECK_HIP(hipMalloc((void **)&done_dev, cells));
    CHECK_AZ(azul_batch_selfplay(b, t, NULL, NULL, action_dev, reward_dev, done_dev, NULL, NULL, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    int32_t *action = malloc(cells * 4), *reward = malloc(cells * 4);
    uint8_t *done = malloc(cells), *records = malloc((size_t)n * AZUL_RECORD_BYTES);
    CHECK_HIP(hipMemcpy(action, action_dev, cells * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(reward, reward_dev, cells * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(done, done_dev, cells, hipMemcpyDeviceToHost));
    CHECK_AZ(azul_batch_get_state(b, 0, n, records, NULL));
    FILE *f = fopen(argv[4], "wb");
    if (!f) { perror(argv[4]); return 4; }
    fwrite(action, 4, cells, f);
    fwrite(reward, 4, cells, f);
    fwrite(done, 1, cells, f);
    fwrite(records, AZUL_RECORD_BYTES, (size_t)n, f);
    fclose(f);
    CHECK_AZ(azul_batch_destroy(b));
    printf("%s: %d games x %d moves written\n", azul_version(), n, t);
    return 0;
}
