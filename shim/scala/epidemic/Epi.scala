// UNCOMPILED HERE: this image has no JVM toolchain.  The file is checked mechanically against include/epi.h by
// tests/test_shim_signatures_cpu.py (every function bound, arity and ValueLayout of every argument, size and field
// offsets of every struct) -- not by a compiler.  Needs Java 22+ (java.lang.foreign) and Scala 3.
//
// Panama FFM binding of libepi.so, the MI355X engine behind EpidemicEnv / WorldToy / Calibrate / DDQNAgent.
// One downcall handle per function of include/epi.h; the struct layouts mirror the C structs field by field.
package epidemic

import java.lang.foreign.*
import java.lang.foreign.ValueLayout.*
import java.lang.invoke.MethodHandle

object Epi {
  private val linker = Linker.nativeLinker()
  private val lib    = SymbolLookup.libraryLookup(sys.env.getOrElse("EPI_LIB", "libepi.so"), Arena.global())

  private def fn(name: String, res: MemoryLayout, args: MemoryLayout*): MethodHandle =
    linker.downcallHandle(lib.find(name).orElseThrow(), FunctionDescriptor.of(res, args*))

  // ---- downcall handles: one per function of include/epi.h, in header order
  val version = fn("epi_version", JAVA_INT)
  val lastError = fn("epi_last_error", JAVA_INT, ADDRESS, JAVA_LONG)
  val lastStatus = fn("epi_last_status", JAVA_INT)
  val deviceCount = fn("epi_device_count", JAVA_INT)
  val setDevice = fn("epi_set_device", JAVA_INT, JAVA_INT)
  val allocInt = fn("epi_alloc_int", JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS)
  val allocIntBatch = fn("epi_alloc_int_batch", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS)
  val envStepBatch = fn("epi_env_step_batch", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT)
  val scaleRow = fn("epi_scale_row", JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_DOUBLE, ADDRESS)
  val gravityMatrix = fn("epi_gravity_matrix", JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_DOUBLE, ADDRESS, JAVA_INT)
  val gravityCsr = fn("epi_gravity_csr", JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_DOUBLE, ADDRESS, ADDRESS, ADDRESS, JAVA_INT)
  val javaGaussians = fn("epi_java_gaussians", JAVA_INT, JAVA_LONG, JAVA_INT, ADDRESS)
  val javaGaussianTable = fn("epi_java_gaussian_table", JAVA_INT, JAVA_LONG, JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT)
  val agentActions = fn("epi_agent_actions", JAVA_INT, ADDRESS, JAVA_INT, ADDRESS)
  val mlpInit = fn("epi_mlp_init", JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS)
  val mlpGreedyBatch = fn("epi_mlp_greedy_batch", JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_DOUBLE, ADDRESS, ADDRESS)
  val gridCreate = fn("epi_grid_create", ADDRESS, ADDRESS)
  val gridRun = fn("epi_grid_run", JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS)
  val gridRunStrided = fn("epi_grid_run_strided", JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS)
  val gridRunNoalloc = fn("epi_grid_run_noalloc", JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val gridStatus = fn("epi_grid_status", JAVA_INT, ADDRESS)
  val gridDestroy = fn("epi_grid_destroy", JAVA_INT, ADDRESS)
  val gridSetTiming = fn("epi_grid_set_timing", JAVA_INT, ADDRESS, JAVA_INT)
  val gridKernelTime = fn("epi_grid_kernel_time", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val gridScreenStats = fn("epi_grid_screen_stats", JAVA_INT, ADDRESS, ADDRESS, ADDRESS)
  val gridArgmin = fn("epi_grid_argmin", JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, ADDRESS)
  val calibrateGrid = fn("epi_calibrate_grid", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val gridMultiCreate = fn("epi_grid_multi_create", ADDRESS, ADDRESS, ADDRESS, JAVA_INT)
  val gridMultiRun = fn("epi_grid_multi_run", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val gridMultiRanks = fn("epi_grid_multi_ranks", JAVA_INT, ADDRESS, ADDRESS, ADDRESS)
  val gridMultiSetTiming = fn("epi_grid_multi_set_timing", JAVA_INT, ADDRESS, JAVA_INT)
  val gridMultiTimes = fn("epi_grid_multi_times", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val gridMultiPeerAccess = fn("epi_grid_multi_peer_access", JAVA_INT, ADDRESS, ADDRESS)
  val gridMultiInjectFailure = fn("epi_grid_multi_inject_failure", JAVA_INT, ADDRESS, JAVA_INT)
  val gridMultiDestroy = fn("epi_grid_multi_destroy", JAVA_INT, ADDRESS)
  val calibrateGridMulti = fn("epi_calibrate_grid_multi", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val worldCreate = fn("epi_world_create", ADDRESS, ADDRESS)
  val worldStep = fn("epi_world_step", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS)
  val worldResultBuffers = fn("epi_world_result_buffers", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val worldStepPacked = fn("epi_world_step_packed", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS)
  val worldPackedBuffer = fn("epi_world_packed_buffer", JAVA_INT, ADDRESS, ADDRESS)
  val worldStepGreedy = fn("epi_world_step_greedy", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS, ADDRESS)
  val worldRolloutGreedy = fn("epi_world_rollout_greedy", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val worldKeepTransitions = fn("epi_world_keep_transitions", JAVA_INT, ADDRESS, JAVA_INT)
  val worldSetMlp = fn("epi_world_set_mlp", JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_DOUBLE)
  val worldGetStates = fn("epi_world_get_states", JAVA_INT, ADDRESS, ADDRESS)
  val worldSetStates = fn("epi_world_set_states", JAVA_INT, ADDRESS, ADDRESS)
  val worldReset = fn("epi_world_reset", JAVA_INT, ADDRESS)
  val worldLastFlows = fn("epi_world_last_flows", JAVA_INT, ADDRESS, ADDRESS, ADDRESS)
  val worldScreenStats = fn("epi_world_screen_stats", JAVA_INT, ADDRESS, ADDRESS, ADDRESS)
  val worldSetTiming = fn("epi_world_set_timing", JAVA_INT, ADDRESS, JAVA_INT)
  val worldKernelTime = fn("epi_world_kernel_time", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val worldDestroy = fn("epi_world_destroy", JAVA_INT, ADDRESS)
  val agentsCreate = fn("epi_agents_create", ADDRESS, JAVA_INT, ADDRESS, JAVA_INT)
  val agentsDestroy = fn("epi_agents_destroy", JAVA_INT, ADDRESS)
  val agentsObserve = fn("epi_agents_observe", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val agentsObserveWorld = fn("epi_agents_observe_world", JAVA_INT, ADDRESS, ADDRESS)
  val agentsReplaySize = fn("epi_agents_replay_size", JAVA_INT, ADDRESS, ADDRESS)
  val agentsSample = fn("epi_agents_sample", JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS)
  val agentsLearn = fn("epi_agents_learn", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val agentsQValues = fn("epi_agents_q_values", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, ADDRESS)
  val agentsGetWeights = fn("epi_agents_get_weights", JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val agentsSetWeights = fn("epi_agents_set_weights", JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  val agentsSetTiming = fn("epi_agents_set_timing", JAVA_INT, ADDRESS, JAVA_INT)
  val agentsKernelTime = fn("epi_agents_kernel_time", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS)

  // ---- struct layouts of include/epi.h (java.lang.foreign inserts no padding: it is spelled out)
  val PROFILE: StructLayout = MemoryLayout.structLayout(   // 72 bytes
    JAVA_DOUBLE.withName("beta_mul"),
    JAVA_DOUBLE.withName("ifr_mul"),
    JAVA_DOUBLE.withName("noise"),
    JAVA_DOUBLE.withName("population_density"),
    JAVA_DOUBLE.withName("gdp_per_capita"),
    JAVA_DOUBLE.withName("healthcare_capacity"),
    JAVA_DOUBLE.withName("age_elderly"),
    JAVA_DOUBLE.withName("connectivity"),
    JAVA_DOUBLE.withName("econ_resilience"))
  val ENV_PARAMS: StructLayout = MemoryLayout.structLayout(   // 48 bytes
    JAVA_DOUBLE.withName("base_beta"),
    JAVA_DOUBLE.withName("base_gamma"),
    JAVA_DOUBLE.withName("base_ifr"),
    JAVA_DOUBLE.withName("noise"),
    JAVA_DOUBLE.withName("reward_clip"),
    JAVA_DOUBLE.withName("risk_factor"))
  val STATE: StructLayout = MemoryLayout.structLayout(   // 64 bytes
    JAVA_DOUBLE.withName("s"),
    JAVA_DOUBLE.withName("i"),
    JAVA_DOUBLE.withName("r"),
    JAVA_DOUBLE.withName("d"),
    JAVA_DOUBLE.withName("v"),
    JAVA_DOUBLE.withName("hosp_cap"),
    JAVA_INT.withName("t"),
    JAVA_INT.withName("t_max"),
    JAVA_INT.withName("seed"),
    JAVA_INT.withName("pad_"))
  val HYPER: StructLayout = MemoryLayout.structLayout(   // 48 bytes
    JAVA_DOUBLE.withName("epsilon_start"),
    JAVA_DOUBLE.withName("epsilon_end"),
    JAVA_INT.withName("epsilon_decay_steps"),
    JAVA_INT.withName("hidden"),
    JAVA_DOUBLE.withName("clip_q"),
    JAVA_DOUBLE.withName("reward_clip"),
    JAVA_INT.withName("batch_size"),
    JAVA_INT.withName("learn_every"))
  val LEARN_HYPER: StructLayout = MemoryLayout.structLayout(   // 56 bytes
    JAVA_DOUBLE.withName("gamma"),
    JAVA_DOUBLE.withName("lr"),
    JAVA_DOUBLE.withName("clip_q"),
    JAVA_DOUBLE.withName("grad_clip"),
    JAVA_DOUBLE.withName("ema_alpha"),
    JAVA_INT.withName("hidden"),
    JAVA_INT.withName("replay_capacity"),
    JAVA_INT.withName("target_update_every"),
    JAVA_INT.withName("max_need"))
  val GRID_DESC: StructLayout = MemoryLayout.structLayout(   // 264 bytes
    JAVA_INT.withName("n"),
    MemoryLayout.paddingLayout(4),
    ADDRESS.withName("pop"),
    ADDRESS.withName("conn"),
    ADDRESS.withName("csr_rowptr"),
    ADDRESS.withName("csr_col"),
    ADDRESS.withName("csr_val"),
    JAVA_INT.withName("seed_idx"),
    JAVA_INT.withName("heterogeneous"),
    STATE.withName("base"),
    ADDRESS.withName("prof"),
    HYPER.withName("hp"),
    JAVA_INT.withName("steps"),
    JAVA_INT.withName("n_beta"),
    ADDRESS.withName("betas"),
    JAVA_INT.withName("n_travel"),
    MemoryLayout.paddingLayout(4),
    ADDRESS.withName("travel"),
    ADDRESS.withName("t_arrive"),
    ADDRESS.withName("has_arrive"),
    ADDRESS.withName("t_peak"),
    ADDRESS.withName("has_peak"),
    JAVA_INT.withName("device"),
    JAVA_INT.withName("pad_"),
    ADDRESS.withName("learn"),
    ADDRESS.withName("learn_intensity"))
  val WORLD_DESC: StructLayout = MemoryLayout.structLayout(   // 200 bytes
    JAVA_INT.withName("n"),
    JAVA_INT.withName("batch"),
    ADDRESS.withName("pop"),
    ADDRESS.withName("conn"),
    ADDRESS.withName("csr_rowptr"),
    ADDRESS.withName("csr_col"),
    ADDRESS.withName("csr_val"),
    JAVA_INT.withName("seed_idx"),
    JAVA_INT.withName("validate"),
    STATE.withName("base"),
    ADDRESS.withName("env"),
    ADDRESS.withName("prof"),
    ADDRESS.withName("env_seeds"),
    HYPER.withName("hp"),
    JAVA_INT.withName("device"),
    JAVA_INT.withName("pad_"))
  val STATE_PACKED: StructLayout = MemoryLayout.structLayout(   // 32 bytes
    JAVA_INT.withName("s"),
    JAVA_INT.withName("i"),
    JAVA_INT.withName("r"),
    JAVA_INT.withName("d"),
    JAVA_INT.withName("v"),
    JAVA_INT.withName("t"),
    JAVA_DOUBLE.withName("hosp_cap"))

  // ---- status handling ---------------------------------------------------------------------------------------
  val OK = 0; val BAD_ARGUMENT = 1; val DEVICE = 2; val LOCAL_MASS = 3; val WORLD_MASS = 4; val ASSERT_STATE = 5
  val INT_RANGE = 6; val UNSUPPORTED = 7

  def lastErrorText(): String = {
    val a = Arena.ofConfined()
    try {
      val buf = a.allocate(1024L)
      lastError.invoke(buf, 1024L)
      buf.getString(0L)
    } finally a.close()
  }

  /** A non-zero status becomes the exception the reference throws at the same place: its `require(...)` failures
    * (WorldToy.scala:94,144; Validation.scala:7-10) are IllegalArgumentException("requirement failed: ..."). */
  def check(status: Int): Unit =
    if (status != OK) {
      val msg = lastErrorText()
      status match {
        case LOCAL_MASS | WORLD_MASS | ASSERT_STATE => throw new IllegalArgumentException(msg)
        case BAD_ARGUMENT                           => throw new IllegalArgumentException(s"epi: $msg")
        case UNSUPPORTED                            => throw new UnsupportedOperationException(s"epi: $msg")
        case _                                      => throw new RuntimeException(s"epi status $status: $msg")
      }
    }

  /** A *_create call returned NULL: raise what it failed with. */
  def checkHandle(h: MemorySegment): MemorySegment = {
    if (h == null || h.address() == 0L) {
      val st = lastStatus.invoke().asInstanceOf[Int]
      check(if (st != OK) st else DEVICE)
    }
    h
  }

  // ---- marshalling helpers -------------------------------------------------------------------------------------
  def doubles(a: Arena, xs: Array[Double]): MemorySegment = a.allocateFrom(JAVA_DOUBLE, xs*)
  def ints(a: Arena, xs: Array[Int]): MemorySegment       = a.allocateFrom(JAVA_INT, xs*)
  def bytes(a: Arena, xs: Array[Byte]): MemorySegment     = a.allocateFrom(JAVA_BYTE, xs*)

  def putState(seg: MemorySegment, index: Long, s: State): Unit = {
    val o = index * STATE.byteSize()
    seg.set(JAVA_DOUBLE, o, s.s); seg.set(JAVA_DOUBLE, o + 8, s.i); seg.set(JAVA_DOUBLE, o + 16, s.r)
    seg.set(JAVA_DOUBLE, o + 24, s.d); seg.set(JAVA_DOUBLE, o + 32, s.v); seg.set(JAVA_DOUBLE, o + 40, s.hospCap)
    seg.set(JAVA_INT, o + 48, s.t); seg.set(JAVA_INT, o + 52, s.tMax); seg.set(JAVA_INT, o + 56, s.seed)
    seg.set(JAVA_INT, o + 60, 0)
  }

  def getState(seg: MemorySegment, index: Long): State = {
    val o = index * STATE.byteSize()
    State(seg.get(JAVA_DOUBLE, o), seg.get(JAVA_DOUBLE, o + 8), seg.get(JAVA_DOUBLE, o + 16), seg.get(JAVA_DOUBLE, o + 24),
          seg.get(JAVA_DOUBLE, o + 32), seg.get(JAVA_DOUBLE, o + 40), seg.get(JAVA_INT, o + 48), seg.get(JAVA_INT, o + 52),
          seg.get(JAVA_INT, o + 56))
  }

  def putProfile(seg: MemorySegment, index: Long, p: Geo.Country): Unit = {
    val o = index * PROFILE.byteSize()
    val v = Array(p.betaMul, p.ifrMul, p.noise, p.populationDensity, p.gdpPerCapita, p.healthcareCapacity,
                  p.ageDistributionElderly, p.connectivityStrength, p.economicResilience)
    var k = 0
    while (k < v.length) { seg.set(JAVA_DOUBLE, o + 8L * k, v(k)); k += 1 }
  }

  def putHyper(seg: MemorySegment, off: Long, hp: HyperParams): Unit = {
    seg.set(JAVA_DOUBLE, off, hp.epsilonStart); seg.set(JAVA_DOUBLE, off + 8, hp.epsilonEnd)
    seg.set(JAVA_INT, off + 16, hp.epsilonDecaySteps); seg.set(JAVA_INT, off + 20, hp.hidden)
    seg.set(JAVA_DOUBLE, off + 24, hp.clipQ); seg.set(JAVA_DOUBLE, off + 32, hp.rewardClip)
    seg.set(JAVA_INT, off + 40, hp.batchSize); seg.set(JAVA_INT, off + 44, hp.learnEvery)
  }

  def putLearnHyper(seg: MemorySegment, index: Long, hp: HyperParams, maxNeed: Int): Unit = {
    val o = index * LEARN_HYPER.byteSize()
    seg.set(JAVA_DOUBLE, o, hp.gamma); seg.set(JAVA_DOUBLE, o + 8, hp.lr); seg.set(JAVA_DOUBLE, o + 16, hp.clipQ)
    seg.set(JAVA_DOUBLE, o + 24, hp.gradClip); seg.set(JAVA_DOUBLE, o + 32, hp.emaAlpha)
    seg.set(JAVA_INT, o + 40, hp.hidden); seg.set(JAVA_INT, o + 44, hp.replayCapacity)
    seg.set(JAVA_INT, o + 48, hp.targetUpdateEvery); seg.set(JAVA_INT, o + 52, maxNeed)
  }

  /** Byte offset of a named field of one of the struct layouts above. */
  def offsetOf(layout: StructLayout, field: String): Long =
    layout.byteOffset(MemoryLayout.PathElement.groupElement(field))
}
