// csi_comm.h -- halo-exchange plans shared by comm.hip and the ABI.
#pragma once
#include "csi_dev.h"

namespace csi {

struct TileInfo {
    int rx = 0, ry = 0, Rx = 1, Ry = 1, periodic_x = 0, periodic_y = 0;
    bool set = false;
};

struct ExSeg {
    FRef f;
    int i0, j0, ni, nj;
    long off;
};
constexpr int MAX_EX_FIELDS = 6;
struct ExPlan {
    int nseg;
    long total;
    ExSeg seg[8 * MAX_EX_FIELDS];
};

int tile_neighbor(const TileInfo& t, int dx, int dy, int xlo, int xhi, int ylo, int yhi);
void build_plan(const GridDev& g, const TileInfo& t, const FRef* fields, int nf, int W, int halo,
                ExPlan& pl, long* dir_off, long* dir_cnt, int* dir_peer);
void launch_pack(const ExPlan& pl, double* buf, int unpack, hipStream_t s);
// peer halo transport: one wave waits until, for every direction d with a neighbour (sync_rank[d] >= 0), the first n[d] flag
// slots of that direction have reached `seq`; sets *err after 3 s instead of hanging
void launch_wait_peers(const unsigned long long* slots, const int* sync_rank, int slots_per_dir, const int* n, unsigned long long seq,
                       unsigned* err, hipStream_t s);

}  // namespace csi
