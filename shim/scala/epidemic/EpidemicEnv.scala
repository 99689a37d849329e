// UNCOMPILED HERE (no JVM toolchain in the build image); replaces src/main/scala/epidemic/env.scala.
//
// Same public surface as the reference's env.scala -- Action, State, EpidemicEnv(...).reset/step/params and
// EpidemicEnv.allocateInt -- with the arithmetic forwarded to libepi.so (include/epi.h).  Callers
// (WorldToy, Trainer, MultiTrainerToy, the ScalaTest specs) compile against it unchanged.
package epidemic

import java.lang.foreign.*
import java.lang.foreign.ValueLayout.*
import scala.util.Random
import Geo.CountryProfiles

enum Action(val id: Int, val label: String):
  case NoneA       extends Action(0, "None")
  case Distancing  extends Action(1, "Distancing")
  case TravelBan   extends Action(2, "TravelBan")
  case Lockdown    extends Action(3, "Lockdown")
  case NormalCare  extends Action(4, "NormalCare")
  case SurgeCare   extends Action(5, "SurgeCare")
  case TargetedVax extends Action(6, "TargetedVax")
  case MassVax     extends Action(7, "MassVax")

object Action:
  val all: Array[Action] = Action.values
  def fromId(i: Int): Action = all(math.max(0, math.min(all.length - 1, i)))

final case class State(s: Double, i: Double, r: Double, d: Double, v: Double, hospCap: Double, t: Int, tMax: Int, seed: Int):
  def n: Double = s + i + r + d
  def util: Double = i / math.max(1.0, hospCap)
  def features: Array[Double] =
    val N = math.max(1.0, n)
    Array(s / N, i / N, r / N, d / N, v / N, util, t.toDouble / math.max(1, tMax))

final class EpidemicEnv(
    baseBeta: Double = 0.35,
    baseGamma: Double = 0.12,
    baseIFR: Double = 0.008,
    noise: Double = 0.05,
    rewardClip: Double = 5.0,
    riskFactor: Double = 0.0,
    countryName: String = "Unknown",
    seed: Int = 42
) {
  // the env's own noise stream stays on the JVM: one nextGaussian() per step, handed to the engine (epi.h,
  // epi_env_step_batch: "gauss[b] is the value rnd.nextGaussian() returns at env.scala:91")
  private val rnd = new Random(seed)
  private val countryProfile: Geo.Country =
    try CountryProfiles.getProfile(countryName)
    catch { case _: Exception => Geo.Country(countryName, 50000000, 0.0, 0.0) }

  // what a WorldToy needs to put this env into an epi_world_desc
  private[epidemic] def profileRecord: Geo.Country = countryProfile
  private[epidemic] def seedValue: Int = seed
  private[epidemic] def putParams(seg: MemorySegment, index: Long): Unit = {
    val o = index * Epi.ENV_PARAMS.byteSize()
    seg.set(JAVA_DOUBLE, o, baseBeta); seg.set(JAVA_DOUBLE, o + 8, baseGamma); seg.set(JAVA_DOUBLE, o + 16, baseIFR)
    seg.set(JAVA_DOUBLE, o + 24, noise); seg.set(JAVA_DOUBLE, o + 32, rewardClip); seg.set(JAVA_DOUBLE, o + 40, riskFactor)
  }

  def reset(s0: State): State = s0

  def step(st: State, act: Action): (State, Double, Boolean) = {
    val a = Arena.ofConfined()
    try {
      val env = a.allocate(Epi.ENV_PARAMS); putParams(env, 0L)
      val prof = a.allocate(Epi.PROFILE); Epi.putProfile(prof, 0L, countryProfile)
      val in = a.allocate(Epi.STATE); Epi.putState(in, 0L, st)
      val action = Epi.ints(a, Array(act.id))
      val gauss = Epi.doubles(a, Array(rnd.nextGaussian()))
      val out = a.allocate(Epi.STATE)
      val reward = a.allocate(JAVA_DOUBLE)
      val done = a.allocate(JAVA_BYTE)
      Epi.check(Epi.envStepBatch.invoke(env, prof, in, action, gauss, out, reward, done, 1).asInstanceOf[Int])
      val next = Epi.getState(out, 0L).copy(tMax = st.tMax, seed = st.seed)
      (next, reward.get(JAVA_DOUBLE, 0L), done.get(JAVA_BYTE, 0L) != 0)
    } finally a.close()
  }

  final case class EnvParams(beta: Double, gamma: Double, ifr: Double, noise: Double)
  def params: EnvParams = EnvParams(baseBeta, baseGamma, baseIFR, noise)
}

object EpidemicEnv {
  def allocateInt(total: Int, weights: Array[Double]): Array[Int] = {
    if (weights.isEmpty) return Array.emptyIntArray
    val a = Arena.ofConfined()
    try {
      val w = Epi.doubles(a, weights)
      val out = a.allocate(JAVA_INT, weights.length.toLong)
      Epi.check(Epi.allocInt.invoke(total, w, weights.length, out).asInstanceOf[Int])
      out.toArray(JAVA_INT)
    } finally a.close()
  }
}
