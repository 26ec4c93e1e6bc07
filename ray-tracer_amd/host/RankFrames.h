#pragma once

// The ranks of one frame inside one process: a thread and a GPU each.  Every rank first does its own
// share of the frame (probe pass, photons, gather), then all of them meet, and only when every share
// came through do they enter the collective that sums the shares.  A rank that throws therefore keeps
// the others out of the collective instead of leaving them waiting in it, and the frame ends with that
// rank's message — what the reference does with a worker's exception: the orchestrating thread rethrows
// it after stopping the others (reference src/Renderer.cpp:511-523,584-595).

#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

class RankFrames
{
public:
    explicit RankFrames(int world) : m_world(world < 1 ? 1 : world), m_failures(static_cast<size_t>(m_world)) {}

    int world() const { return m_world; }

    // A frame that is being given up (a rank failed, or the caller's progress callback said stop):
    // ranks look at this between the pieces of their photon share and trace no further piece.
    void cancel()
    {
        std::lock_guard<std::mutex> lock(m_mutex);
        m_abandoned = true;
    }
    bool abandoned() const
    {
        std::lock_guard<std::mutex> lock(m_mutex);
        return m_abandoned;
    }

    // A rank's report after its share, before the collective (empty text: the share is done).  Returns
    // once every rank has reported; true when none of them failed.
    bool meet(int rank, const std::string& failure)
    {
        std::unique_lock<std::mutex> lock(m_mutex);
        if (!failure.empty())
        {
            m_failures[static_cast<size_t>(rank)] = failure;
            m_failed = true;
            m_abandoned = true;
        }
        m_arrived += 1;
        if (m_arrived == m_world)
        {
            m_everyoneArrived.notify_all();
        }
        else
        {
            m_everyoneArrived.wait(lock, [this]() { return m_arrived == m_world; });
        }
        return !m_failed;
    }

    // A failure inside the collective itself (after a meeting without failures).
    void collectiveFailed(int rank, const std::string& failure)
    {
        std::lock_guard<std::mutex> lock(m_mutex);
        if (m_failures[static_cast<size_t>(rank)].empty()) m_failures[static_cast<size_t>(rank)] = failure;
        m_failed = true;
    }

    // The first failing rank's message (lowest rank), empty when none failed.
    std::string failure() const
    {
        std::lock_guard<std::mutex> lock(m_mutex);
        for (int rank = 0; rank < m_world; ++rank)
        {
            const std::string& text = m_failures[static_cast<size_t>(rank)];
            if (!text.empty())
            {
                return rank == 0 ? text : "another GPU's share of the frame failed (rank " + std::to_string(rank) + "): " + text;
            }
        }
        return std::string();
    }

    // work(rank) for every rank — rank 0 on the calling thread, so that its callbacks fire there —, the
    // meeting, then collective(rank) for every rank iff no share failed.  All threads have ended when
    // this returns; the result is failure().
    std::string run(const std::function<void(int)>& work, const std::function<void(int)>& collective)
    {
        auto rankBody = [&](int rank) {
            std::string failed;
            try
            {
                work(rank);
            }
            catch (const std::exception& e)
            {
                failed = e.what()[0] != '\0' ? e.what() : "unknown failure";
            }
            catch (...)
            {
                failed = "unknown failure";
            }
            if (!meet(rank, failed))
            {
                return;
            }
            try
            {
                collective(rank);
            }
            catch (const std::exception& e)
            {
                collectiveFailed(rank, e.what()[0] != '\0' ? e.what() : "unknown failure");
            }
            catch (...)
            {
                collectiveFailed(rank, "unknown failure");
            }
        };
        std::vector<std::thread> threads;
        for (int rank = 1; rank < m_world; ++rank)
        {
            threads.emplace_back(rankBody, rank);
        }
        rankBody(0);
        for (std::thread& t : threads)
        {
            t.join();
        }
        return failure();
    }

private:
    const int m_world;
    mutable std::mutex m_mutex;
    std::condition_variable m_everyoneArrived;
    int m_arrived = 0;
    bool m_failed = false, m_abandoned = false;
    std::vector<std::string> m_failures;
};
