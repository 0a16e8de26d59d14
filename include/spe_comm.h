/* spe_comm.h - C ABI of libspe_comm.so: the collective layer of the SPE data-parallel hot path over RCCL (xGMI).
 *
 * The reference has no native code; its only parallelism is PyTorch DistributedDataParallel over NCCL.  Each entry
 * replaces the reference Python site cited with it (paths relative to MingXiangL/SPE):
 *
 *   spe_comm_unique_id / spe_comm_init / spe_comm_destroy
 *       util/misc.py:414-436  init_distributed_mode(): torch.distributed.init_process_group(backend='nccl',
 *                             init_method='env://', world_size, rank) + barrier - one process per GPU, rank from the
 *                             launcher's environment.  Rank 0 draws the 128-byte unique id and hands it to the other
 *                             ranks through any side channel (spe_amd/comm.py uses a torch TCPStore on
 *                             MASTER_ADDR:MASTER_PORT); every rank then calls spe_comm_init on ITS current HIP device.
 *   spe_comm_allreduce
 *       main.py:172           DistributedDataParallel(model, ...): the bucketed gradient all-reduce (sum; the division
 *                             by the world size is folded into spe_adamw_flat's grad_scale);
 *       models/conditional_detr.py:438-440   torch.distributed.all_reduce(num_boxes) inside SetCriterion.forward;
 *       util/misc.py:139-163  reduce_dict(): all-reduce of the stacked loss scalars for logging.
 *   spe_comm_broadcast
 *       main.py:172           DistributedDataParallel's construction-time broadcast of rank 0's parameters and buffers.
 *
 * Conventions (as spe_hip.h): plain device pointers and element counts, caller-owned memory, asynchronous on `stream`
 * (a hipStream_t; the caller orders it against its compute stream with events), status-code returns: 0 ok, -1 not
 * initialised, -2 invalid argument, otherwise 1000 + ncclResult_t or a hipError_t.  One communicator per process.
 */
#ifndef SPE_COMM_H
#define SPE_COMM_H
#ifdef __cplusplus
extern "C" {
#endif

typedef void* spe_stream_t;

#define SPE_COMM_ID_BYTES 128
#define SPE_COMM_F32 0
#define SPE_COMM_BF16 1

int spe_comm_unique_id(void* id_out);
int spe_comm_init(int rank, int world, const void* id);
int spe_comm_world(int* rank, int* world);
int spe_comm_allreduce(void* buf, long count, int dtype, spe_stream_t stream);
int spe_comm_broadcast(void* buf, long count, int dtype, int root, spe_stream_t stream);
int spe_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif
