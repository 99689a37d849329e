// UNCOMPILED HERE (no JVM toolchain in the build image); replaces src/main/scala/epidemic/Calibrate.scala.
//
// Same Targets and gridSurface(...) as the reference; the whole beta x travel sweep -- every rollout, its arrival /
// peak metrics, the two RMSE terms and the loss -- runs on the GPU(s) in one call (include/epi.h: epi_calibrate_grid,
// or epi_calibrate_grid_multi over the devices named in EPI_DEVICES, e.g. "0,1,2,3,4,5,6,7").
package epidemic

import java.lang.foreign.*
import java.lang.foreign.ValueLayout.*
import Validation._

object Calibrate {
  final case class Targets(
      tArrive: Map[String, Double],
      peakI: Map[String, Double],
      threshI: Double = 10.0
  )

  def gridSurface(
      hp: HyperParams,
      cfg: TrainConfig,
      base: State,
      seedIdx: Int,
      targets: Targets,
      betaGrid: Seq[Double],
      travelScales: Seq[Double],
      validate: Boolean = true,
      heterogeneous: Boolean = true
  ): (Seq[(Double, Double, Double)], (Double, (Double, Double))) = {
    val countries = ToyConn.C.map(_.name)
    val n = countries.size
    val bestPair0 = (betaGrid.headOption.getOrElse(0.25), travelScales.headOption.getOrElse(1.0))
    if (betaGrid.isEmpty || travelScales.isEmpty) return (Seq.empty, (Double.MaxValue, bestPair0))
    val a = Arena.ofConfined()
    try {
      val d = a.allocate(Epi.GRID_DESC)
      d.fill(0.toByte)
      def at(f: String): Long = Epi.offsetOf(Epi.GRID_DESC, f)
      d.set(JAVA_INT, at("n"), n)
      d.set(ADDRESS, at("pop"), Epi.ints(a, ToyConn.C.map(_.pop).toArray))
      d.set(ADDRESS, at("conn"), Epi.doubles(a, ToyConn.W.flatten))          // scaleRow per travel scale happens in the engine
      d.set(JAVA_INT, at("seed_idx"), seedIdx)
      d.set(JAVA_INT, at("heterogeneous"), if (heterogeneous) 1 else 0)
      Epi.putState(d.asSlice(at("base"), Epi.STATE.byteSize()), 0L, base)
      val prof = a.allocate(Epi.PROFILE, n.toLong)
      countries.zipWithIndex.foreach { case (c, k) => Epi.putProfile(prof, k.toLong, Geo.CountryProfiles.getProfile(c)) }
      d.set(ADDRESS, at("prof"), prof)
      Epi.putHyper(d, at("hp"), hp)
      d.set(JAVA_INT, at("steps"), cfg.stepsPerEpoch)
      d.set(JAVA_INT, at("n_beta"), betaGrid.size); d.set(ADDRESS, at("betas"), Epi.doubles(a, betaGrid.toArray))
      d.set(JAVA_INT, at("n_travel"), travelScales.size); d.set(ADDRESS, at("travel"), Epi.doubles(a, travelScales.toArray))
      // Targets maps as value + presence arrays: a missing key means "no target for this country" (getOrElse on the
      // measured value makes its term zero, Calibrate.scala:104,107)
      d.set(ADDRESS, at("t_arrive"), Epi.doubles(a, countries.map(c => targets.tArrive.getOrElse(c, 0.0)).toArray))
      d.set(ADDRESS, at("has_arrive"), Epi.bytes(a, countries.map(c => (if (targets.tArrive.contains(c)) 1 else 0).toByte).toArray))
      d.set(ADDRESS, at("t_peak"), Epi.doubles(a, countries.map(c => targets.peakI.getOrElse(c, 0.0)).toArray))
      d.set(ADDRESS, at("has_peak"), Epi.bytes(a, countries.map(c => (if (targets.peakI.contains(c)) 1 else 0).toByte).toArray))
      // rollouts long enough for their agents to learn before an argmax-Q step (WorldToy.scala:147-167)
      val learn = a.allocate(Epi.LEARN_HYPER)
      Epi.putLearnHyper(learn, 0L, hp, hp.batchSize * 2)
      d.set(ADDRESS, at("learn"), learn)
      d.set(ADDRESS, at("learn_intensity"), Epi.ints(a, countries.map(WorldToy.learningIntensity).toArray))

      val g = betaGrid.size * travelScales.size
      val loss = a.allocate(JAVA_DOUBLE, g.toLong)
      val bestIdx = a.allocate(JAVA_INT); val bestLoss = a.allocate(JAVA_DOUBLE)
      val devices = sys.env.get("EPI_DEVICES").map(_.split(",").map(_.trim.toInt)).getOrElse(Array(WorldToy.device))
      if (devices.length == 1) {
        d.set(JAVA_INT, at("device"), devices(0))
        Epi.check(Epi.calibrateGrid.invoke(d, loss, MemorySegment.NULL, MemorySegment.NULL, bestIdx, bestLoss).asInstanceOf[Int])
      } else
        Epi.check(Epi.calibrateGridMulti.invoke(d, Epi.ints(a, devices), devices.length, loss, MemorySegment.NULL, MemorySegment.NULL,
                                                bestIdx, bestLoss).asInstanceOf[Int])
      val l = loss.toArray(JAVA_DOUBLE)
      val rows = for ((beta, b) <- betaGrid.zipWithIndex; (ts, k) <- travelScales.zipWithIndex) yield (beta, ts, l(b * travelScales.size + k))
      val best = bestLoss.get(JAVA_DOUBLE, 0L)
      // strict '<' from Double.MaxValue: when no loss is below it the pair stays the first grid point (:96-97,112)
      val pair = if (best < Double.MaxValue) { val r = rows(bestIdx.get(JAVA_INT, 0L)); (r._1, r._2) } else bestPair0
      (rows.toSeq, (best, pair))
    } finally a.close()
  }
}
