/*
 * mrl_envs.h -- C ABI of the MI355X-native batched RL-environment step engine.
 *
 * One shared library (libmrl_envs.so, built by hipcc for gfx950) replaces, for
 * the step hot path only, what the reference exposes through its nanobind
 * modules.  Every entry point cites the reference interface it stands in for
 * (paths under /root/reference).  Plain C types only: no torch, no C++ classes.
 *
 *   reference                                         this ABI
 *   ------------------------------------------------  ---------------------------
 *   OvercookedSimulator.__init__                      mrl_overcooked_create
 *     src/overcooked_env/bindings.cpp:14-71,
 *     Manager::Impl::init mgr.cpp:138-188
 *   HanabiSimulator.__init__                          mrl_hanabi_create
 *     src/hanabi_env/bindings.cpp:10-36, mgr.cpp:144-166
 *   CartpoleSimulator.__init__                        mrl_cartpole_create
 *     src/cartpole_env/bindings.cpp:11-24, mgr.cpp:138-163
 *   Manager::step (all three)                         mrl_step
 *     src/overcooked_env/mgr.cpp:196-199,
 *     src/hanabi_env/mgr.cpp:174-177, src/cartpole_env/mgr.cpp:169-172
 *   Manager::*Tensor()  -> madrona::py::Tensor        mrl_tensor(slot)
 *     src/overcooked_env/mgr.cpp:201-259,
 *     src/hanabi_env/mgr.cpp:179-232, src/cartpole_env/mgr.cpp:174-202
 *   Manager::~Manager                                 mrl_destroy
 *   FATAL()/abort on error (mgr.cpp:177)              return code + mrl_last_error
 *
 * Ownership: the simulator owns every buffer it exports for its whole
 * lifetime (the reference's Manager owns its export buffers the same way,
 * mgr.hpp:63); mrl_tensor hands out device pointers that never move.  The
 * caller writes actions in place into the ACTION tensor before mrl_step and
 * reads results in place after it, exactly as the reference's Python wrappers
 * do (envs/overcooked_env.py:104-113, pantheonrl_extension/vectorenv.py:306-329).
 *
 * Streams: mrl_step only enqueues work on the HIP stream it is given (NULL =
 * the default stream) and never synchronises the host; ordering with the
 * caller's own work on that stream is the stream's.  A handle is bound to one
 * device and is not thread-safe (neither is the reference's Manager).
 *
 * New functionality with no reference counterpart (the reference is
 * single-device): the two-phase step and the episode
 * base used to keep episode numbering identical when worlds are sharded over
 * several GPUs.
 */
#ifndef MRL_ENVS_H
#define MRL_ENVS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRL_ABI_VERSION 4

/* return codes */
enum {
    MRL_OK = 0,
    MRL_ERR_INVALID = 1, /* bad argument / unsupported configuration          */
    MRL_ERR_DEVICE = 2,  /* no usable gfx950 device, or a HIP call failed      */
    MRL_ERR_SLOT = 3     /* tensor slot not exported by this game             */
};

/* element types of exported tensors (madrona::py::Tensor::ElementType subset) */
enum { MRL_INT8 = 0, MRL_UINT8 = 1, MRL_INT32 = 2, MRL_FLOAT32 = 3, MRL_UINT32 = 4 };

enum { MRL_GAME_OVERCOOKED = 1, MRL_GAME_HANABI = 2, MRL_GAME_CARTPOLE = 3, MRL_GAME_SIMPLECOOKED = 4, MRL_GAME_BALANCE = 5 };

typedef struct mrl_sim mrl_sim;

#define MRL_MAX_DIMS 6
typedef struct mrl_tensor_desc {
    void *data;                    /* device pointer (HBM), valid until mrl_destroy   */
    int32_t dtype;                 /* MRL_INT8 ...                                    */
    int32_t ndim;
    int64_t shape[MRL_MAX_DIMS];
    int64_t strides[MRL_MAX_DIMS]; /* in elements; exported views may be strided      */
    int32_t device;                /* HIP device ordinal                              */
    int32_t reserved;
} mrl_tensor_desc;

/* ------------------------------------------------------------------ */
/* Overcooked  (reference: src/overcooked_env)                          */
/* ------------------------------------------------------------------ */

/* Mirrors Manager::Config (mgr.hpp:16-35): the Python layout transform
 * (envs/overcooked_env.py:261-371) produces exactly these fields.  Limits are
 * the reference's: height*width <= 255 (WorldState.size is uint8, sim.hpp:86),
 * num_players <= 64 (sim.hpp:14); reward / recipe entries are stored as uint8
 * (sim.hpp:95-99). */
typedef struct mrl_overcooked_config {
    int64_t height, width, num_players;
    int64_t placement_in_pot_rew, dish_pickup_rew, soup_pickup_rew;
    int64_t horizon;
    const int64_t *terrain;        /* height*width, row-major, TerrainT values 0..6 */
    const int64_t *start_player_x; /* num_players                                   */
    const int64_t *start_player_y; /* num_players                                   */
    const int64_t *recipe_values;  /* 16, index 4*onions + tomatoes                 */
    const int64_t *recipe_times;   /* 16                                            */
} mrl_overcooked_config;

/* Tensor slots = ExportID (sim.hpp:23-37) + this engine's world-major views.
 * P players, N worlds, C = height*width, F = 5P + 16.
 *   DONE              int32 (N)
 *   ACTIVE_AGENT      int32 (P, N)            all ones (sim.cpp:619)
 *   ACTION            int32 (P, N, 1)         written by the caller
 *   OBSERVATION       int8  (P*C, N, F)       strided view of OBS_WORLD_MAJOR; row id
 *                                             p*C + cell like LocationXID (sim.cpp:633).
 *                                             The reference pads rows to 336 bytes
 *                                             (sim.hpp:125-127) and callers slice [:F];
 *                                             here the last dimension is F already.
 *   ACTION_MASK       int32 (P, N, 6)         all ones (sim.cpp:616-618)
 *   REWARD            int32 (P, N)
 *   WORLD_ID          int32 (P, N)            [p, n] = n
 *   AGENT_ID          int32 (P, N)            [p, n] = p
 *   LOCATION_WORLD_ID int32 (P*C, N)          [r, n] = n
 *   LOCATION_ID       int32 (P*C, N)          [r, n] = r
 *   OBS_WORLD_MAJOR   int8  (N, P, H, W, F)   contiguous; what the kernel writes
 *   STATE_PLAYERS     uint8 (N, P, 8)         pos, orientation, pad, pad, held{name,on,tom,tick}
 *   STATE_OBJECTS     uint8 (N, C, 4)         name, onions, tomatoes, cooking_tick
 *   STATE_TIMESTEP    int32 (N)
 */
enum {
    MRL_OVERCOOKED_DONE = 0,
    MRL_OVERCOOKED_ACTIVE_AGENT = 1,
    MRL_OVERCOOKED_ACTION = 2,
    MRL_OVERCOOKED_OBSERVATION = 3,
    MRL_OVERCOOKED_ACTION_MASK = 4,
    MRL_OVERCOOKED_REWARD = 5,
    MRL_OVERCOOKED_WORLD_ID = 6,
    MRL_OVERCOOKED_AGENT_ID = 7,
    MRL_OVERCOOKED_LOCATION_WORLD_ID = 8,
    MRL_OVERCOOKED_LOCATION_ID = 9,
    MRL_OVERCOOKED_OBS_WORLD_MAJOR = 10,
    MRL_OVERCOOKED_STATE_PLAYERS = 11,
    MRL_OVERCOOKED_STATE_OBJECTS = 12,
    MRL_OVERCOOKED_STATE_TIMESTEP = 13
};

/* replaces OvercookedSimulator(exec_mode=CUDA, gpu_id, num_worlds, **layout)
 * (bindings.cpp:14-71).  There is no CPU execution mode behind this ABI. */
int mrl_overcooked_create(const mrl_overcooked_config *cfg, int gpu_id, uint32_t num_worlds, mrl_sim **out);

/* ------------------------------------------------------------------ */
/* Simplecooked  (reference: src/overcooked2_env, the world the trainer */
/* and the Colab notebook use: train/env_utils.py:3)                    */
/* ------------------------------------------------------------------ */

/* replaces SimplecookedSimulator(exec_mode=CUDA, gpu_id, num_worlds, **layout)
 * (src/overcooked2_env/bindings.cpp:15-71; Manager::Config mgr.hpp:16-35 has the fields of
 * mrl_overcooked_config, so the struct is shared).  Differences from Overcooked that the caller sees:
 * terrain values follow overcooked2's enum AIR, POT, COUNTER, ONION_SOURCE, DISH_SOURCE, SERVING,
 * TOMATO_SOURCE (sim.hpp:40); height*width <= 100 and num_players <= 2 (sim.hpp:12-13); rows are
 * F = 5P + 10 bytes (sim.hpp:121-123), so OBSERVATION is int8 (P*C, N, 5P+10) and OBS_WORLD_MAJOR
 * (N, P, H, W, 5P+10); dish_pickup_rew is paid (sim.cpp:241-246).  Slots: the MRL_OVERCOOKED_* ids
 * (ExportID is the same list, sim.hpp:22-37) plus STATE_DISHES_OUT int32 (N) = WorldState.num_dishes_out. */
enum { MRL_SIMPLECOOKED_STATE_DISHES_OUT = 14 };
int mrl_simplecooked_create(const mrl_overcooked_config *cfg, int gpu_id, uint32_t num_worlds, mrl_sim **out);

/* ------------------------------------------------------------------ */
/* Hanabi  (reference: src/hanabi_env; 2 players, hand of 5)            */
/* ------------------------------------------------------------------ */

typedef struct mrl_hanabi_config {
    uint32_t colors, ranks, players, max_information_tokens, max_life_tokens;
} mrl_hanabi_config;

#define MRL_HANABI_OBS_SIZE 658   /* OBS_SIZE   sim.hpp:29 */
#define MRL_HANABI_STATE_SIZE 783 /* STATE_SIZE sim.hpp:30 */
#define MRL_HANABI_NUM_MOVES 20   /* NUM_MOVES  sim.hpp:19 */

/* Slots = ExportID (src/hanabi_env/sim.hpp:38-49); shapes as mgr.cpp:179-232:
 *   DONE int32 (N); ACTIVE_AGENT int32 (2,N); ACTION int32 (2,N,1);
 *   OBSERVATION int8 (2,N,658); ACTION_MASK int32 (2,N,20); REWARD float32 (2,N);
 *   WORLD_ID / AGENT_ID int32 (2,N); STATE int8 (2,N,783);
 *   OBSERVATION, STATE and ACTION_MASK are strided views (see the strides in the
 *   descriptor) into one array of 896-byte blocks [state 784 | mask 80 | pad 32],
 *   world-major with the two agents of a world back to back: one step writes whole
 *   cache lines only.  OBSERVATION is the first 658 bytes of the STATE row -- the same
 *   memory: the reference fills the state by copying the observation and appending the
 *   own hand (generateObsState, sim.cpp:367-379), so the two tensors agree on those
 *   bytes by construction, always (also for the agent whose buffers stay stale: both
 *   are refreshed together), and this engine writes them once.  For a configuration
 *   smaller than the full game (fewer colours / tokens) the observation has fewer than
 *   658 entries and OBSERVATION is exported exactly that wide, (2, N, obs_size): the row
 *   continues with the agent's own hand, which is hidden from the observer (the reference
 *   declares 658 entries whatever the configuration; its wrappers read [:obs_size],
 *   envs/hanabi_env.py:92-104).
 *   GAME uint8 (N, 176): the raw per-world game record (tests only; layout in
 *   csrc/hanabi.hip); RESET_COUNT uint32 (1): worlds that finished in the last completed step (written by
 *   phase 2); SHARD_COUNT uint32 (1): worlds that finished in the last mrl_step_phase1 -- what the ranks of a
 *   sharded batch all-gather between the phases (mrl_step_phase2_gathered); mrl_step does not update it. */
enum {
    MRL_HANABI_DONE = 0,
    MRL_HANABI_ACTIVE_AGENT = 1,
    MRL_HANABI_ACTION = 2,
    MRL_HANABI_OBSERVATION = 3,
    MRL_HANABI_ACTION_MASK = 4,
    MRL_HANABI_REWARD = 5,
    MRL_HANABI_WORLD_ID = 6,
    MRL_HANABI_AGENT_ID = 7,
    MRL_HANABI_STATE = 8,
    MRL_HANABI_GAME = 9,
    MRL_HANABI_RESET_COUNT = 10,
    MRL_HANABI_SCAN_TIMEOUT = 11, /* uint32 (1): see mrl_step */
    MRL_HANABI_SHARD_COUNT = 12
};

int mrl_hanabi_create(const mrl_hanabi_config *cfg, int gpu_id, uint32_t num_worlds, mrl_sim **out);

/* ------------------------------------------------------------------ */
/* Cartpole  (reference: src/cartpole_env)                              */
/* ------------------------------------------------------------------ */

/* Slots = ExportID (src/cartpole_env/sim.hpp:17-24); shapes as mgr.cpp:174-202:
 *   RESET int32 (N,1); ACTION int32 (N,1); STATE float32 (N,4) (the Python
 *   binding calls it observation_tensor, bindings.cpp:28); REWARD float32 (N,1);
 *   WORLD_ID int32 (N,1); RESET_COUNT uint32 (1); SHARD_COUNT uint32 (1) (both as for Hanabi). */
enum {
    MRL_CARTPOLE_RESET = 0,
    MRL_CARTPOLE_ACTION = 1,
    MRL_CARTPOLE_STATE = 2,
    MRL_CARTPOLE_REWARD = 3,
    MRL_CARTPOLE_WORLD_ID = 4,
    MRL_CARTPOLE_RESET_COUNT = 5,
    MRL_CARTPOLE_SCAN_TIMEOUT = 6, /* uint32 (1): see mrl_step */
    MRL_CARTPOLE_SHARD_COUNT = 7
};

int mrl_cartpole_create(int gpu_id, uint32_t num_worlds, mrl_sim **out);

/* ------------------------------------------------------------------ */
/* Balance beam  (reference: src/balance_beam_env)                      */
/* ------------------------------------------------------------------ */

/* Slots = ExportID (src/balance_beam_env/sim.hpp:22-32); shapes as mgr.cpp:177-223:
 *   DONE int32 (N); ACTIVE_AGENT int32 (2,N) all ones; ACTION int32 (2,N,1), values 0..3 = moves -2,-1,+1,+2;
 *   OBSERVATION int32 (2,N,7) = own position history x[0..2], partner's x[3..5] (positions + 2), steps left;
 *   ACTION_MASK int32 (2,N,4) all ones; REWARD float32 (2,N); WORLD_ID / AGENT_ID int32 (2,N);
 *   RESET_COUNT uint32 (1); SHARD_COUNT uint32 (1) (both as for Hanabi).  The observation IS the world state (sim.cpp:99-112).  Episodes are numbered
 *   in ascending world order like Cartpole's; mrl_step = phase 1 + phase 2. */
enum {
    MRL_BALANCE_DONE = 0,
    MRL_BALANCE_ACTIVE_AGENT = 1,
    MRL_BALANCE_ACTION = 2,
    MRL_BALANCE_OBSERVATION = 3,
    MRL_BALANCE_ACTION_MASK = 4,
    MRL_BALANCE_REWARD = 5,
    MRL_BALANCE_WORLD_ID = 6,
    MRL_BALANCE_AGENT_ID = 7,
    MRL_BALANCE_RESET_COUNT = 8,
    MRL_BALANCE_SHARD_COUNT = 9
};

/* replaces BalanceBeamSimulator(exec_mode=CUDA, gpu_id, num_worlds) (src/balance_beam_env/bindings.cpp:10-24) */
int mrl_balance_create(int gpu_id, uint32_t num_worlds, mrl_sim **out);

/* ------------------------------------------------------------------ */
/* Common                                                               */
/* ------------------------------------------------------------------ */

/* One environment step for every world: replaces Manager::step.
 * Hanabi, Cartpole and the balance beam number new episodes in ascending world order, which
 * takes a prefix sum over the worlds that finished.  Where the kernel exists (Cartpole up to 4 M
 * worlds, Hanabi up to 262144) mrl_step is ONE launch: every workgroup publishes its count of
 * finishing worlds and one of its waves looks back at the lower workgroups' counts while the others
 * already stream out results.  The look-back never depends on another workgroup making progress: a
 * count that has not appeared after a short bounded wait is recounted by the waiting wave from that
 * workgroup's inputs (csrc/episode_scan.hpp), so no dispatch order or co-residency is assumed.
 * Otherwise, for the sharded path and under mrl_debug_set("fused_step", 2), it is two launches
 * (phase 1, phase 2: the kernel boundary is the grid-wide hand-off).  The persistent multi-step
 * launches of mrl_rollout_random DO wait for other workgroups inside the kernel; they need all
 * their workgroups resident and are therefore launched cooperatively.  Those waits are
 * bounded; if one ever expired the SCAN_TIMEOUT tensor of the game becomes nonzero, the
 * episode numbers from that step on are unspecified, and every later mrl_step* /
 * mrl_rollout_random on the simulator returns MRL_ERR_DEVICE (the host learns it from a
 * word in mapped host memory: no device call, no sync). */
int mrl_step(mrl_sim *sim, void *hip_stream);

/* HIP graphs.  Every mrl_step* / mrl_rollout_random / mrl_step_sequence only enqueues kernels on the stream it is given, so
 * Overcooked and Simplecooked calls can be captured (hipStreamBeginCapture, torch.cuda.graph) and replayed as they are.
 * Hanabi, Cartpole and the balance beam keep launch-to-launch state on the host by default (which half of the
 * double-buffered episode counter is current, the epoch of the single-launch step's look-back) and pass it in kernel
 * arguments; captured as such, a replay would run with stale values, so those calls return MRL_ERR_INVALID while the stream
 * is capturing.  mrl_prepare_graph_capture(sim, stream) -- called once, OUTSIDE a capture; it synchronises the stream --
 * moves that state into device memory for the rest of the simulator's life: every step then enqueues a one-thread launch
 * that advances it in front of its kernels, which read it from there, and captured steps replay correctly (a replay of K
 * captured steps is K more steps).  The price is that extra launch per step (~2 us of GPU time), also outside graphs, and
 * mrl_rollout_random runs one launch per step on such a simulator (its persistent form is a cooperative launch, which
 * cannot be captured).  No-op for Overcooked and Simplecooked. */
int mrl_prepare_graph_capture(mrl_sim *sim, void *hip_stream);

/* Launch shape of the simulator's step kernel: out = {workgroups, threads per workgroup, LDS bytes per
 * workgroup, worlds per wavefront (0 where that is not how the game is mapped)}.  For DESIGN.md's
 * occupancy arithmetic and the tests that guard it; no reference counterpart. */
int mrl_launch_shape(const mrl_sim *sim, uint32_t out[4]);

/* 1 if an in-kernel wait of an earlier call expired (see mrl_step), else 0.  Reads host memory only. */
int mrl_scan_timed_out(const mrl_sim *sim);

/* Same step, but actions are read from caller memory instead of the ACTION
 * tensor (same dtype/shape/layout, device pointer).  Saves the copy the
 * reference wrappers make (static_actions.copy_, envs/overcooked_env.py:107). */
int mrl_step_with_actions(mrl_sim *sim, const int32_t *actions_dev, void *hip_stream);

/* Several simulators stepped by ONE launch: any mix of Overcooked layouts, sizes, player counts and world counts on one device
 * (the reference's Config holds a single terrain, src/overcooked_env/sim.hpp:44-57, so a curriculum over several layouts is
 * several simulators there, each with a step call of its own).  The grid is the concatenation of the simulators' grids and
 * every workgroup runs its simulator's step on that simulator's parameters, which travel in the kernel arguments.  Same
 * results as one mrl_step_with_actions per simulator (actions_dev_or_null == NULL or an entry NULL: that simulator's
 * ACTION tensor).  At most 8 simulators per call; the generic step kernel is used, so large single-layout batches are
 * better off with their own, specialised launch -- this is for many small sub-batches, where the launches are what
 * costs.  Overcooked only (MRL_ERR_INVALID otherwise, and for the few-worlds-of-a-large-layout configurations whose
 * workgroups share one state copy). */
int mrl_step_many(mrl_sim *const *sims, uint32_t count, const int32_t *const *actions_dev_or_null, void *hip_stream);

/* The same with the caller's actions as int64 (same shape and layout): what the reference's harness hands
 * `env.n_step` (scripts/overcooked_example.py:99-106: `torch.randint_like` of a long tensor), which the reference
 * wrapper narrows with a gather + copy kernel per step (envs/overcooked_env.py:104-107).  Here the step kernel reads
 * the 8-byte values itself and mirrors them into the ACTION tensor, so the wrapped step is one launch.  Overcooked and
 * Simplecooked; MRL_ERR_INVALID for the other games. */
int mrl_step_with_actions_i64(mrl_sim *sim, const int64_t *actions_dev, void *hip_stream);

/* Two-phase step for world batches sharded over several GPUs (Hanabi and
 * Cartpole draw each new episode's seed from one global counter,
 * src/hanabi_env/sim.cpp:449-451, src/cartpole_env/sim.cpp:51-53):
 *   phase 1 = transition + termination test, leaves the number of finishing
 *             worlds of this shard in SHARD_COUNT (a one-workgroup launch behind the
 *             step kernel that adds up its per-workgroup counts);
 *   phase 2 = re-seed and reset the finishing worlds, taking episode indices
 *             episode_base, episode_base+1, ... in ascending world order.
 * episode_base_dev is a device pointer to one uint32 (the caller computes it
 * from the gathered RESET_COUNTs of the lower ranks without a host sync);
 * NULL means "use and advance the simulator's own counter" (single GPU).
 * mrl_step gives the same results as phase 1 + phase 2(NULL).  Overcooked has no episode counter:
 * phase 1 is the whole step and phase 2 is a no-op. */
int mrl_step_phase1(mrl_sim *sim, const int32_t *actions_dev_or_null, void *hip_stream);
int mrl_step_phase2(mrl_sim *sim, const uint32_t *episode_base_dev, void *hip_stream);

/* Phase 2 for rank `rank` of `num_ranks` (1..1024) with the exchange left on the device: counts_dev holds the
 * SHARD_COUNT word of every rank for this step, in rank order -- exactly what one all-gather of the SHARD_COUNT
 * tensors delivers.  The re-seeding launch itself adds the lower ranks' counts to the simulator's own episode
 * counter to get its base and advances that counter by the sum over all ranks, so a sharded step is
 * phase 1 -> all-gather of one word per rank -> this call, with no other device work in between and no host
 * sync.  The counter must have been set by mrl_reseed_shard.  No-op for games without an episode counter. */
int mrl_step_phase2_gathered(mrl_sim *sim, const uint32_t *counts_dev, uint32_t num_ranks, uint32_t rank, void *hip_stream);

/* The same exchange WITHOUT a collective (round 4): a device-side mailbox.  Every rank owns a small block of device memory;
 * mrl_exchange_create allocates it for this rank of `num_ranks` (<= MRL_MAX_RANKS) and returns its IPC handle
 * (MRL_IPC_HANDLE_BYTES bytes, hipIpcGetMemHandle); the caller gives every rank the handles of all ranks, in rank order (one
 * all-gather of 64 bytes at set-up), and mrl_exchange_connect maps the peers' blocks (hipIpcOpenMemHandle).  From then on
 *     mrl_step_exchanged(sim, actions_or_null, stream)
 * is one whole step of the shard: phase 1; a one-workgroup launch that adds up the shard's finished worlds and stores
 * (step tag, count) into word `rank` of every rank's mailbox -- num_ranks stores over xGMI --; phase 2, whose
 * workgroups poll the num_ranks words of their own mailbox for this step's tag and number the episodes as
 * mrl_step_phase2_gathered does.  No host call and no collective between the launches, so a captured or free-running
 * loop needs no rendezvous; every rank must call it the same number of times (a rank that never publishes leaves the others'
 * phase 2 polling until the bounded wait expires: SCAN_TIMEOUT, as for the persistent rollouts).  One process per rank (a
 * rank's own handle is not opened).  Reference: one process-wide atomic, src/hanabi_env/sim.cpp:449-451. */
#define MRL_MAX_RANKS 16
#define MRL_IPC_HANDLE_BYTES 64
int mrl_exchange_create(mrl_sim *sim, uint32_t num_ranks, uint32_t rank, uint8_t *ipc_handle_out);
int mrl_exchange_connect(mrl_sim *sim, const uint8_t *ipc_handles_of_all_ranks);
int mrl_step_exchanged(mrl_sim *sim, const int32_t *actions_dev_or_null, void *hip_stream);

/* Rollout-buffer side of a trainer.  The reference's MAPPO loop clones the observation
 * and state tensors after every step and copies them into the buffer slot of that step
 * (train/MAPPO/main_player.py:245-247, utils/shared_buffer.py:115 chooseinsert).  Here the caller hands the step the
 * slot instead: from this call on every mrl_step* / mrl_rollout_random / mrl_step_sequence of the simulator writes
 * its observation slab -- int8 (N, P, H, W, F), the OBS_WORLD_MAJOR layout, `bytes` = N*P*H*W*F -- to obs_dev_or_null
 * and leaves the OBSERVATION / OBS_WORLD_MAJOR tensors untouched; NULL hands the output back to them.  Same bytes, same
 * stores, no copy: the kernels take the slab's address from their launch arguments and never read it back.  The
 * buffer must stay valid until the work enqueued before the next call of this function has finished.  A buffer that
 * starts on a 16-byte boundary costs nothing; one that does not (slot k of a dense (T, N, P, H, W, F) buffer whose
 * N*P*H*W*F is not a multiple of 16) is accepted too and STAGED: the kernels stream 16-byte chunks from an aligned base,
 * so the step writes a slab of the simulator's own (allocated at the first such call; the exported tensors stay
 * untouched) and one device-to-device copy behind the launch, on the same stream, moves it to the slot -- one more
 * pass over the slab per step.  Overcooked and Simplecooked (MRL_ERR_INVALID for the other games).  Host-only call
 * (but for that allocation): nothing is enqueued. */
int mrl_set_observation_output(mrl_sim *sim, void *obs_dev_or_null, uint64_t bytes);

/* The same for a whole rollout buffer: a ring of num_slots observation slots, slot s at base + s * slot_stride_bytes
 * (>= N*P*H*W*F; with base and stride multiples of 16 the slots are written in place, otherwise they are staged as
 * above and the multi-step launches run one launch + one copy per step).  Step number k counted from this call -- whether it is a launch of its own or step k of a
 * multi-step launch (mrl_rollout_random, mrl_step_sequence: the kernel moves on to the next slot itself) -- writes its
 * observations to slot k % num_slots.  One mrl_rollout_random(sim, T, ...) then fills a T-slot buffer with T steps of
 * random-policy experience in one launch.  base == NULL hands the output back to the simulator's own tensor; one slot is
 * mrl_set_observation_output.  The count lives on the host: a launch captured in a HIP graph keeps the slot(s) it was
 * captured with.  Overcooked and Simplecooked. */
int mrl_set_observation_ring(mrl_sim *sim, void *base_dev_or_null, uint64_t slot_stride_bytes, uint32_t num_slots);

/* Sets the simulator's own episode counter (next index handed out). Sharded
 * runs call it once after create with the global world offset semantics the
 * caller wants; it does not touch world state. */
int mrl_set_episode_counter(mrl_sim *sim, uint32_t next_episode, void *hip_stream);

/* Re-initialises world i of this shard as global world (world_offset + i) of a
 * num_worlds_total batch: episode index world_offset + i, counter starts at
 * num_worlds_total -- what a single simulator of the whole batch would hold
 * after construction.  No-op for Overcooked (its initial state is not seeded). */
int mrl_reseed_shard(mrl_sim *sim, uint32_t world_offset, uint32_t num_worlds_total, void *hip_stream);

/* num_steps steps driven by an open-loop action array: actions_dev holds num_steps consecutive
 * ACTION tensors (step k at offset k * elements(ACTION), same dtype and layout, device pointer).
 * Same results as num_steps calls of mrl_step_with_actions.  Overcooked layouts whose observation
 * slab fits the LDS tile run them in ONE launch with the worlds' state resident in LDS (every
 * step still writes its observations, rewards and dones); everything else is one launch per step. */
int mrl_step_sequence(mrl_sim *sim, const int32_t *actions_dev, uint32_t num_steps, void *hip_stream);

/* The random policies of the reference's benchmark harnesses, drawn on the device
 * (SURVEY.md section 8f item 1): num_steps environment steps with no action tensor to fill.
 *   Overcooked  randint(high=6) per agent            scripts/overcooked_example.py:99-106
 *   Cartpole    randint(high=2)                      scripts/cartpole_example.py:53-87
 *   Simplecooked randint(high=6) per agent; Balance beam randint(high=4) per agent (their example scripts)
 *   Hanabi      argmax(rand * mask), i.e. a uniformly random legal move of the player
 *               to move                              scripts/hanabi_example.py:53-82
 * All draws come from one counter-based hash of (seed, step index k = first_step,
 * first_step+1, ..., world w, player q), so a stream can be replayed through mrl_step:
 *     h = lo32(seed) ^ k*0x9E3779B9 ^ w*0x85EBCA6B ^ (q+1)*0xC2B2AE35 ^ hi32(seed)*0x27D4EB2F
 *     h ^= h>>16; h *= 0x7FEB352D; h ^= h>>15; h *= 0x846CA68B; h ^= h>>16;   (all mod 2^32)
 *   Overcooked  action = (h * 6) >> 32
 *   Cartpole    action = h >> 31                      (q = 0)
 *   Hanabi      action = position of the j-th set bit of the mover's 20-bit legal-move
 *               mask, j = (h * popcount(mask)) >> 32  (q = the mover)
 * Every step writes its outputs exactly like mrl_step; afterwards the ACTION tensor holds
 * the last step's draws (Hanabi: the mover's entry).  Overcooked layouts whose observation
 * slab fits the LDS tile, and Hanabi / Cartpole batches whose workgroups all fit the GPU at
 * once (65536 Hanabi worlds, 1 M Cartpole worlds do), run all num_steps in ONE launch with
 * the worlds' state resident in LDS / registers; otherwise it is one launch per step. */
int mrl_rollout_random(mrl_sim *sim, uint32_t num_steps, uint64_t seed, uint32_t first_step, void *hip_stream);

int mrl_tensor(mrl_sim *sim, int slot, mrl_tensor_desc *out);
int mrl_game(const mrl_sim *sim);
uint32_t mrl_num_worlds(const mrl_sim *sim);
/* name of the dominant kernel of this simulator's step, as rocprofv3 prints it */
const char *mrl_kernel_name(const mrl_sim *sim);
/* name of the kernel the next mrl_rollout_random will run: a persistent one ("mrl_hanabi_rollout", "mrl_cartpole_rollout":
 * all steps in one cooperative launch) while the device can hold the whole grid at once, otherwise mrl_kernel_name's, once per
 * step.  A cooperative launch the runtime refuses switches the simulator to the latter for good; tests and benchmarks read
 * this to know which of the two they are looking at. */
const char *mrl_rollout_kernel_name(const mrl_sim *sim);
/* algorithmic HBM bytes one step moves per world */
uint64_t mrl_bytes_per_world_step(const mrl_sim *sim);
void mrl_destroy(mrl_sim *sim);

/* Test and measurement knobs, consulted by every LATER mrl_*_create until they are forgotten again (the library reads no
 * environment variable).  They are process-global: set them, create, forget them -- the Python binding's `debug_knobs`
 * context manager does exactly that, also when the create throws.  Keys (with their meanings: csrc/capi.hip, kDebugKeys):
 * overcooked.wpw, overcooked.whole_max, overcooked.lds_max, overcooked.share_max_players, overcooked.share_private,
 * overcooked.no_share, overcooked.lds_pad, overcooked.no_fixed, overcooked.no_direct, overcooked.whole_store,
 * overcooked.store_policy, overcooked.wide_rollout, overcooked.groups, overcooked.shared_consts, overcooked.variant,
 * hanabi.variant, hanabi.pairing, hanabi.no_persistent, cartpole.no_persistent, cartpole.persistent_max, cartpole.variant, fused_step (0 the library's choice, 1 one launch,
 * 2 two launches), fused_heal_test, inject_scan_timeout, and (diagnostic build) ablate, stamps.  key == NULL forgets all of
 * them.  Unknown key: MRL_ERR_INVALID.  No reference counterpart (the reference has MADRONA_* environment variables for its
 * JIT cache only). */
int mrl_debug_set(const char *key, int64_t value);

/* Measurement aid for bench.py's roofline.peak_measured: one float4 stream over caller buffers on
 * device gpu_id, enqueued on hip_stream.  mode 0: copy src -> dst (reads + writes bytes each);
 * mode 1: fill dst, plain stores; mode 2: fill dst, write-through (sc1) stores like the step
 * kernels' observation stream.  bytes: multiple of 16, buffers 16-byte aligned.  No reference counterpart. */
int mrl_probe_stream(void *dst_dev, const void *src_dev, uint64_t bytes, int mode, int gpu_id, void *hip_stream);

/* message of the last failing call on this thread ("" if none) */
const char *mrl_last_error(void);
int mrl_abi_version(void);
/* Which sources this binary was built from: the first 16 hex digits of a sha256 over the .hip and .hpp files of csrc/, its
 * Makefile and this header (the Makefile computes it and compiles it in; the Python binding computes the same value from the files
 * lying beside the library and refuses a library whose hash differs).  Measurements that cannot be taken inside a benchmark
 * run (PMC traffic, profiles/step_traffic.json) are keyed by it.  No reference counterpart. */
const char *mrl_build_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* MRL_ENVS_H */
