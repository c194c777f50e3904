// csi_hostgroup.h -- host-channel tile group: the ranks are PROCESSES (one context each) that talk through a POSIX
// shared-memory segment and HIP IPC instead of an RCCL communicator (csi_comm_init_host, include/csi.h).
//
// Why it exists: RCCL refuses two ranks on one device, and the boxes this library is developed on have one GPU.  The
// in-process tile group (csi_local_group) runs real decompositions there, but inside ONE address space: it can hand a neighbour's
// array to the peer halo transport as a plain pointer.  Between processes nothing of that is true -- arrays and flag words
// reach a neighbour only as (hipIpcGetMemHandle of the allocation, offset) pairs opened with hipIpcOpenMemHandle, exactly what
// one process per GPU does on a real node.  This group carries the two small collectives of the peer set-up and the grouped
// send / receive of the halo exchange over the host, so that the WHOLE multi-process path of the peer transport (IPC mapping of
// caller-owned and library-owned arrays, write-through stores and flags into another process's memory, the error words) runs
// on a one-GPU box (tests/test_gpu_hostgroup.py).  Host-synchronous like the in-process group: built for correctness, not speed.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace csi {

struct HostGroup;      // opaque (csi_hostgroup.hip)

// Join (rank 0 .. world - 1 all call it with the same name).  Rank 0 OWNS the name: it unlinks whatever an earlier run left under it and
// creates the segment with O_EXCL, stamped with its pid; the other ranks never create -- they accept an existing segment only if it is
// complete, initialised, for this world size and stamped by a LIVING process (kill(pid, 0): the ranks must share a PID namespace, which
// processes sharing /dev/shm on one node normally do), else they drop it and look again until the time-out.  Everybody waits for every
// rank; the name is unlinked once all have it mapped and on every failure path of rank 0.
HostGroup* hostgroup_join(const char* shm_name, int world, int rank, std::string* err);
void hostgroup_leave(HostGroup* g);
int hostgroup_world(const HostGroup* g);
// all ranks: rank r's `nb` bytes end up in out[r * nb ...] everywhere (collectives are called in the same order by every rank)
bool hostgroup_allgather(HostGroup* g, const void* mine, size_t nb, std::vector<uint8_t>& out, std::string* err);
// before this rank's send buffer is packed again (or freed): every message posted from it has been copied out
bool hostgroup_wait_consumed(HostGroup* g, std::string* err);
// this rank's send buffer was (re)allocated: export it (a hipMalloc allocation, `bytes` long)
bool hostgroup_set_sendbuf(HostGroup* g, void* dev_ptr, size_t bytes, std::string* err);
// the grouped send / receive of a halo exchange (eight directions; peers < 0 or counts 0: none): offsets and counts in doubles
// into the send buffer announced with hostgroup_set_sendbuf / into recvbuf.  Messages between a pair of ranks match in posting
// order (RCCL's rule).  The stream is synchronised before posting (the pack kernel has filled the buffer) and after the copies.
bool hostgroup_sendrecv(HostGroup* g, hipStream_t stream, double* recvbuf, const long* soff, const long* scnt, const int* speer,
                        const long* roff, const long* rcnt, const int* rpeer, std::string* err);

}  // namespace csi
