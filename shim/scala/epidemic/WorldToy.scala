// UNCOMPILED HERE (no JVM toolchain in the build image); replaces src/main/scala/epidemic/WorldToy.scala.
//
// Same constructor, `nodes`, stepOne / stepOneGreedy and snapshot / restore / reset as the reference's WorldToy, so
// MultiTrainerToy, Calibrate and the specs keep calling it unchanged.  The world itself -- local dynamics, mobility
// apportionment, the two mass checks -- is an epi_world on the GPU (include/epi.h); what stays here is the order of
// the calls: every node's agent chooses its action, the world steps, the agents observe, the nodes that are due learn.
package epidemic

import java.lang.foreign.*
import java.lang.foreign.ValueLayout.*
import scala.util.Random
import epidemic.Validation.*

final class WorldToy(
    hp: HyperParams,
    base: State,
    seedIdx: Int = 0,
    rng: Random = new Random(7),
    conn: Array[Array[Double]] = ToyConn.W,
    makeEnv: () => EpidemicEnv = () => new EpidemicEnv(),
    makeEnvFor: Option[(String, Int) => EpidemicEnv] = None,
    validate: Boolean = false,
    parallelism: Int = Runtime.getRuntime.availableProcessors(),
    countryHyperParams: Map[String, HyperParams] = Map.empty
) extends AutoCloseable {

  private var gStep: Long = 0L
  private val n = ToyConn.C.size
  private val arena = Arena.ofShared()

  final case class Node(
      idx: Int,
      name: String,
      pop: Int,
      env: EpidemicEnv,
      agent: DDQNAgent,
      hyperParams: HyperParams,
      var s: State,
      var lastA: Int = 0,
      var lastLearnStep: Long = 0L
  )

  private val nodeHps: Vector[HyperParams] = ToyConn.C.map(c => countryHyperParams.getOrElse(c.name, hp)).toVector
  // one engine handle for all nodes' agents: their transitions are filed from device memory and the nodes that are
  // due on a step learn in one call
  private val pool = new AgentPool(nodeHps, maxGradAccum = 4, device = WorldToy.device)
  private val envs: Vector[EpidemicEnv] =
    ToyConn.C.map(c => makeEnvFor.map(f => f(c.name, c.pop)).getOrElse(makeEnv())).toVector

  // ---- the device world (epi_world_create): graph, per-country env parameters, profiles and noise seeds ---------
  private val world: MemorySegment = {
    val a = Arena.ofConfined()
    try {
      val d = a.allocate(Epi.WORLD_DESC)
      d.fill(0.toByte)
      def at(f: String): Long = Epi.offsetOf(Epi.WORLD_DESC, f)
      d.set(JAVA_INT, at("n"), n); d.set(JAVA_INT, at("batch"), 1)
      d.set(ADDRESS, at("pop"), Epi.ints(a, ToyConn.C.map(_.pop).toArray))
      d.set(ADDRESS, at("conn"), Epi.doubles(a, conn.flatten))
      d.set(JAVA_INT, at("seed_idx"), seedIdx); d.set(JAVA_INT, at("validate"), if (validate) 1 else 0)
      Epi.putState(d.asSlice(at("base"), Epi.STATE.byteSize()), 0L, base)
      val env = a.allocate(Epi.ENV_PARAMS, n.toLong); val prof = a.allocate(Epi.PROFILE, n.toLong)
      var c = 0
      while (c < n) { envs(c).putParams(env, c.toLong); Epi.putProfile(prof, c.toLong, envs(c).profileRecord); c += 1 }
      d.set(ADDRESS, at("env"), env); d.set(ADDRESS, at("prof"), prof)
      d.set(ADDRESS, at("env_seeds"), Epi.ints(a, envs.map(_.seedValue).toArray))   // env.scala:42: the engine draws each env's Random(seed) stream
      Epi.putHyper(d, at("hp"), hp)
      d.set(JAVA_INT, at("device"), WorldToy.device)
      val h = Epi.checkHandle(Epi.worldCreate.invoke(d).asInstanceOf[MemorySegment])
      Epi.check(Epi.worldKeepTransitions.invoke(h, 1).asInstanceOf[Int])
      h
    } finally a.close()
  }

  // page-locked result arrays of the handle, read in place after every step
  private val (outStates, outRewards) = {
    val a = Arena.ofConfined()
    try {
      val ps = a.allocate(ADDRESS); val pr = a.allocate(ADDRESS)
      Epi.check(Epi.worldResultBuffers.invoke(world, ps, pr, MemorySegment.NULL).asInstanceOf[Int])
      (ps.get(ADDRESS, 0L).reinterpret(Epi.STATE.byteSize() * n), pr.get(ADDRESS, 0L).reinterpret(8L * n))
    } finally a.close()
  }
  private val actionsIn = arena.allocate(JAVA_INT, n.toLong)

  private def readStates(): Vector[State] = {
    val a = Arena.ofConfined()
    try {
      val buf = a.allocate(Epi.STATE, n.toLong)
      Epi.check(Epi.worldGetStates.invoke(world, buf).asInstanceOf[Int])
      Vector.tabulate(n)(c => Epi.getState(buf, c.toLong))
    } finally a.close()
  }

  private val initStates: Vector[State] = readStates()   // WorldToy.scala:28-32, formed by the engine

  val nodes: Vector[Node] = ToyConn.C.zipWithIndex.map { case (cfg, i) =>
    val agent = new DDQNAgent(nodeHps(i), 7, Action.values.length, pool, i)
    val s0 = initStates(i)
    if (validate) assertState(s0, Invariants(cfg.pop, base.tMax))
    val p = envs(i).params
    val r0 = if (p.gamma > 0) p.beta / p.gamma else 0.0
    WandB.log(Map(
      s"env_beta_${cfg.name}" -> p.beta,
      s"env_ifr_${cfg.name}" -> p.ifr,
      s"env_noise_${cfg.name}" -> p.noise,
      s"env_R0_${cfg.name}" -> r0
    ))
    Node(i, cfg.name, cfg.pop, envs(i), agent, nodeHps(i), s0, 0, 0L)
  }.toVector

  def snapshotStates(): Vector[State] = nodes.map(_.s)

  def restoreStates(ss: Vector[State]): Unit = {
    val a = Arena.ofConfined()
    try {
      val buf = a.allocate(Epi.STATE, n.toLong)
      ss.zipWithIndex.foreach { case (s, c) => Epi.putState(buf, c.toLong, s) }
      Epi.check(Epi.worldSetStates.invoke(world, buf).asInstanceOf[Int])
    } finally a.close()
    nodes.zip(ss).foreach { case (nd, s) => nd.s = s }
  }

  def resetToInitial(): Unit = {
    Epi.check(Epi.worldReset.invoke(world).asInstanceOf[Int])
    nodes.indices.foreach(i => nodes(i).s = initStates(i))
  }

  private def collect(): Vector[(String, State, Double)] =
    Vector.tabulate(n) { c =>
      val s3 = Epi.getState(outStates, c.toLong)
      nodes(c).s = s3
      nodes(c).lastA = actionsIn.get(JAVA_INT, 4L * c)
      (nodes(c).name, s3, outRewards.get(JAVA_DOUBLE, 8L * c))
    }

  def stepOne(tMax: Int): Vector[(String, State, Double)] = {
    // every node's agent chooses (WorldToy.scala:88); act() draws from the agent's own Random(9) and, on its greedy
    // branch, asks its network on the device
    var c = 0
    while (c < n) { actionsIn.set(JAVA_INT, 4L * c, nodes(c).agent.act(nodes(c).s)); c += 1 }
    // env.step for every node, mobility, both mass checks (a failed check comes back as IllegalArgumentException)
    Epi.check(Epi.worldStep.invoke(world, actionsIn, MemorySegment.NULL, tMax, outStates, outRewards).asInstanceOf[Int])
    // agent.observe(Transition(s, a, r, s2, done)) for every node (:99), from the planes the world kept on the device
    Epi.check(Epi.agentsObserveWorld.invoke(pool.handle, world).asInstanceOf[Int])
    val out = collect()
    gStep += 1
    // each country learns on its own schedule (:150-167); the nodes due on this step share one call
    val sizes = pool.replaySizes()
    val due = nodes.filter { nd =>
      gStep - nd.lastLearnStep >= nd.hyperParams.learnEvery && sizes(nd.idx) >= nd.hyperParams.batchSize * 4
    }
    if (due.nonEmpty) {
      pool.learn(due.map(_.idx).toArray, due.map(_.hyperParams.batchSize).toArray,
                 due.map(nd => WorldToy.learningIntensity(nd.name)).toArray, due.map(_.agent.stepCount).toArray)
      due.foreach(_.lastLearnStep = gStep)
    }
    out
  }

  def stepOneGreedy(tMax: Int): Vector[(String, State, Double)] = {
    var c = 0
    while (c < n) { actionsIn.set(JAVA_INT, 4L * c, nodes(c).agent.actGreedy(nodes(c).s)); c += 1 }
    Epi.check(Epi.worldStep.invoke(world, actionsIn, MemorySegment.NULL, tMax, outStates, outRewards).asInstanceOf[Int])
    collect()
  }

  /** The trainer's evaluation loop (MutliTrainer.scala:128-161) in one call: `steps` greedy steps on the device, every
    * node asking its own agent's network; returns the I series [step][country]. */
  def rolloutGreedy(steps: Int, tMax: Int): Array[Array[Double]] = {
    val a = Arena.ofConfined()
    try {
      val series = a.allocate(JAVA_DOUBLE, steps.toLong * n)
      val finalStates = a.allocate(Epi.STATE, n.toLong)
      Epi.check(Epi.worldRolloutGreedy.invoke(world, pool.handle, steps, tMax, series, MemorySegment.NULL, MemorySegment.NULL,
                                              finalStates).asInstanceOf[Int])
      nodes.indices.foreach(c => nodes(c).s = Epi.getState(finalStates, c.toLong))
      series.toArray(JAVA_DOUBLE).grouped(n).toArray
    } finally a.close()
  }

  override def close(): Unit = { Epi.worldDestroy.invoke(world); pool.close(); arena.close() }
}

object WorldToy {
  /** HIP device the worlds and their agents live on (EPI_DEVICE, default 0). */
  val device: Int = sys.env.get("EPI_DEVICE").map(_.toInt).getOrElse(0)

  /** gradAccumSteps of a country's learn call (WorldToy.scala:157-162). */
  def learningIntensity(name: String): Int = name match {
    case "India" => 2
    case _       => 1
  }
}
