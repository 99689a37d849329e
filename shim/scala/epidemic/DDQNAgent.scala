// UNCOMPILED HERE (no JVM toolchain in the build image); replaces src/main/scala/epidemic/DDQNAgent.scala
// (MLP.scala and ReplayBuffer.scala are then no longer used by the agent: networks and replay buffers live on the GPU).
//
// Same public surface as the reference's DDQNAgent.  What stays on the JVM is what decides WHICH action source is
// used: the step counter, epsilon and the agent's Random(9) stream.  Networks, replay buffer and the learn step are
// the engine's (include/epi.h, epi_agents_*).  Agents of one WorldToy share one epi_agents handle (AgentPool), so
// that the world can file their transitions from device memory and let all nodes that are due learn in one call.
package epidemic

import java.lang.foreign.*
import java.lang.foreign.ValueLayout.*

/** One epi_agents handle: `hps.length` independent DDQN learners on one device. */
final class AgentPool(hps: IndexedSeq[HyperParams], maxGradAccum: Int = 4, device: Int = 0) extends AutoCloseable {
  val size: Int = hps.length
  private val arena = Arena.ofShared()
  val handle: MemorySegment = {
    val rec = arena.allocate(Epi.LEARN_HYPER, size.toLong)
    var k = 0
    while (k < size) { Epi.putLearnHyper(rec, k.toLong, hps(k), hps(k).batchSize * math.max(1, maxGradAccum)); k += 1 }
    Epi.checkHandle(Epi.agentsCreate.invoke(size, rec, device).asInstanceOf[MemorySegment])
  }
  private val ema = Array.fill(size)(0.0)
  def emaOf(idx: Int): Double = ema(idx)

  /** DDQNAgent.learn for several agents in one set of launches; returns their losses. */
  def learn(ids: Array[Int], batch: Array[Int], gradAccum: Array[Int], steps: Array[Long]): Array[Double] = {
    val a = Arena.ofConfined()
    try {
      val loss = a.allocate(JAVA_DOUBLE, ids.length.toLong)
      val emaOut = a.allocate(JAVA_DOUBLE, ids.length.toLong)
      Epi.check(Epi.agentsLearn.invoke(handle, Epi.ints(a, ids), ids.length, Epi.ints(a, batch), Epi.ints(a, gradAccum),
                                       a.allocateFrom(JAVA_LONG, steps*), loss, emaOut).asInstanceOf[Int])
      val e = emaOut.toArray(JAVA_DOUBLE)
      var k = 0
      while (k < ids.length) { ema(ids(k)) = e(k); k += 1 }
      loss.toArray(JAVA_DOUBLE)
    } finally a.close()
  }

  def replaySizes(): Array[Int] = {
    val a = Arena.ofConfined()
    try {
      val out = a.allocate(JAVA_INT, size.toLong)
      Epi.check(Epi.agentsReplaySize.invoke(handle, out).asInstanceOf[Int])
      out.toArray(JAVA_INT)
    } finally a.close()
  }

  /** q[8] of one state on one agent's online network, and the first argmax. */
  def qValues(idx: Int, s: State): (Array[Double], Int) = {
    val a = Arena.ofConfined()
    try {
      val st = a.allocate(Epi.STATE); Epi.putState(st, 0L, s)
      val q = a.allocate(JAVA_DOUBLE, 8L)
      val g = a.allocate(JAVA_INT)
      Epi.check(Epi.agentsQValues.invoke(handle, Epi.ints(a, Array(idx)), 1, st, 0, q, g).asInstanceOf[Int])
      (q.toArray(JAVA_DOUBLE), g.get(JAVA_INT, 0L))
    } finally a.close()
  }

  def observe(idx: Int, tr: Transition): Unit = {
    val a = Arena.ofConfined()
    try {
      val s = a.allocate(Epi.STATE); Epi.putState(s, 0L, tr.s)
      val s2 = a.allocate(Epi.STATE); Epi.putState(s2, 0L, tr.s2)
      Epi.check(Epi.agentsObserve.invoke(handle, Epi.ints(a, Array(idx)), 1, s, Epi.ints(a, Array(tr.a)), Epi.doubles(a, Array(tr.r)),
                                         s2, Epi.bytes(a, Array((if (tr.done) 1 else 0).toByte))).asInstanceOf[Int])
    } finally a.close()
  }

  override def close(): Unit = { Epi.agentsDestroy.invoke(handle); arena.close() }
}

final class DDQNAgent private[epidemic] (hp: HyperParams, stateSize: Int, actionSize: Int, pool: AgentPool, idx: Int) {

  /** The reference's constructor: an agent with a pool of its own. */
  def this(hp: HyperParams, stateSize: Int, actionSize: Int) = this(hp, stateSize, actionSize, new AgentPool(Vector(hp)), 0)

  require(stateSize == 7 && actionSize == 8, "the engine's networks are 7 -> hidden -> hidden -> 8 (State.features, Action.values)")

  private var steps: Long = 0L
  private val rnd = new scala.util.Random(9)

  private def epsilon: Double = {
    val frac = math.min(1.0, steps.toDouble / math.max(1, hp.epsilonDecaySteps))
    hp.epsilonStart + (hp.epsilonEnd - hp.epsilonStart) * frac
  }

  private[epidemic] def stepCount: Long = steps
  private[epidemic] def poolIndex: Int = idx
  private[epidemic] def agentPool: AgentPool = pool

  def actGreedy(s: State): Int = pool.qValues(idx, s)._2

  def actWithEpsilon(s: State, eps: Double, countStep: Boolean = false): Int = {
    if (countStep) steps += 1
    if (rnd.nextDouble() < eps) rnd.nextInt(actionSize)
    else pool.qValues(idx, s)._2
  }

  def act(s: State): Int = actWithEpsilon(s, epsilon, countStep = true)

  def observe(tr: Transition): Unit = pool.observe(idx, tr)

  def learn(batch: Int, gradAccumSteps: Int = 4): Double =
    pool.learn(Array(idx), Array(batch), Array(gradAccumSteps), Array(steps))(0)

  def epsilonNow: Double = epsilon
  def emaLossNow: Double = pool.emaOf(idx)
  def replaySize: Int = pool.replaySizes()(idx)
  def qValues(s: State): Array[Double] = pool.qValues(idx, s)._1
}
