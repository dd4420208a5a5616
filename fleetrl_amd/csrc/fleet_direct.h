// Direct AQL submission of a run of single-step launches (host side only; fleet_direct.hip).  Internal to the library.
#pragma once
#include <string>
#include <vector>

#include "fleet_device.h"

struct FleetDirect;  // an HSA queue of the handle's own on the HIP device's agent + the step kernels' code object loaded through HSA

// opens the queue, loads `<library>.gfx950.hsaco` from beside the library.  FLEET_OK / FLEET_ERR_*; *err says why not.
int fleet_direct_open(int hip_device, FleetDirect** out, std::string* err);
void fleet_direct_close(FleetDirect* q);
// the launch every packet of the next runs repeats, and the tape its action pointers walk: `tape_len` rows of `row_bytes` from `tape`
// (argument blocks in device memory, one per tape row; blocks until they are uploaded).  No run may be in flight.
// `split`: cover the grid with two ranges of workgroups on two queues (large batches; fleet_direct.hip)
int fleet_direct_prepare(FleetDirect* q, const FleetStepLaunch& launch, const void* tape, int tape_len, size_t row_bytes, bool split,
                         std::string* err);
int fleet_direct_parts(FleetDirect* q);  // 1 or 2: how the prepared run is laid out
// `steps` launches (tape rows 0, 1, ... cyclically), asynchronous.  Fences: every packet acquires at agent scope (the vector / scalar
// L1s are invalidated) and releases NOTHING -- except the last, which releases at system scope; `timed`: the first and the last packet
// of the run carry completion signals with dispatch timestamps (fleet_direct_wait reports their span).
int fleet_direct_submit(FleetDirect* q, int steps, bool timed, std::string* err);
bool fleet_direct_busy(FleetDirect* q);  // a run is in flight
// waits for everything submitted; spans_us (nullable): first-start -> last-end of every timed run since the last wait, in order
int fleet_direct_wait(FleetDirect* q, std::vector<double>* spans_us, std::string* err);
