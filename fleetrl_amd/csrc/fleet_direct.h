// Direct AQL submission of single-step launches (host side only; fleet_direct.hip).  Internal to the library.
#pragma once
#include <string>
#include <vector>

#include "fleet_device.h"

struct FleetDirect;  // HSA queues of the handle's own on the HIP device's agent + the step kernels' code object loaded through HSA

// Opens the queue, loads `<library>.gfx950.hsaco` from beside the library, checks that it is the library's twin (source hash) and
// PROBES the placement the mode relies on (workgroup w of every launch on the same die).  FLEET_OK / FLEET_ERR_*; *err says why not.
// FLEET_ERR_UNSUPPORTED: the platform does not place workgroups the way the mode needs (callers fall back to HIP's launches).
int fleet_direct_open(int hip_device, FleetDirect** out, std::string* err);
void fleet_direct_close(FleetDirect* q);
// What the probe found: map[k] = the die of workgroups w with (w & 7) == k on queue 0 (-1: no probe yet); *num_xcc = dies of the agent;
// *any_grid = the map also held across grids that are not multiples of 8 workgroups
int fleet_direct_probed(FleetDirect* q, int map[8], int* num_xcc, int* any_grid);
// How a grid is laid over the queues: 1 = the whole grid on queue 0; 2 = two ranges of workgroups, part_grid[0] a multiple of 8 and
// small enough for the kernel's packed first-workgroup field (16 bits).  A pure function (testable without a device).
int fleet_direct_plan(unsigned grid, bool split, unsigned part_grid[2]);
// the launch every packet of the next submissions repeats, and the tape its action pointers walk: `tape_len` rows of `row_bytes` from
// `tape` (argument blocks in device memory, one per tape row; blocks until they are uploaded).  No launch may be in flight.
// `split`: cover the grid with two ranges of workgroups on two queues (large batches; fleet_direct.hip).
// Nothing of the handle's prepared state changes unless the call succeeds.
int fleet_direct_prepare(FleetDirect* q, const FleetStepLaunch& launch, const void* tape, int tape_len, size_t row_bytes, bool split,
                         std::string* err);
int fleet_direct_parts(FleetDirect* q);  // 1 or 2: how the prepared launch is laid out
// `steps` launches (tape rows 0, 1, ... cyclically) behind the run's placement-record launch, asynchronous.  Fences: every packet acquires
// at agent scope (the vector / scalar L1s are invalidated; the first at system scope) and releases NOTHING -- except the last, which
// releases at system scope.
// `timed`: completion signals with dispatch timestamps -- 1: on the first and the last packet of the run (fleet_direct_wait reports the
// span); 2: on every packet (fleet_direct_wait reports each launch's own start -> end).
int fleet_direct_submit(FleetDirect* q, int steps, int timed, std::string* err);
bool fleet_direct_busy(FleetDirect* q);   // something submitted has not completed
// waits for everything submitted; spans_us (nullable): per timed run since the last wait, in order (see fleet_direct_submit)
int fleet_direct_wait(FleetDirect* q, std::vector<double>* spans_us, std::string* err);
// Test hook (negative tests of the placement guard) --
// kind 1: the NEXT run's placement record is shifted by one workgroup (what its launches would see if the queue's first die moved in
//         the middle of a run);
// kind 2: in the uploaded argument block of tape row `tape_row`, the first workgroup of the grid shifted by one (every workgroup then
//         steps its neighbour's envs).
int fleet_direct_fault(FleetDirect* q, int kind, int tape_row, std::string* err);
