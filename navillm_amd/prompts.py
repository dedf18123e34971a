"""Prompt templates of the agents, byte-exact (every entry of tests/golden/g6_prompts.json, which holds the strings the
reference's own `get_*_prompt` methods return): tasks/agents/r2r.py, reverie.py, soon.py, cvdn.py (navigation /
object_grounding / summarization / embodied_qa) and tasks/agents/llava.py:13-17 (3dqa).  Between agents only the task
sentences differ; the skeleton is shared."""

_HIST = "Following is the History, which contains the visual information of your previous decisions.\n"
_NAV_TASK = {
    "r2r": "### Instruction: Navigate following the instruction. {} \n",
    "reverie": "### Instruction: Go to the location to complete the given task. Task: {} \n",
    "soon": "### Instruction: Find the described target. Target: {} \n",
    "cvdn": "### Instruction: Find the described room according the given dialog. Target: {} \n",
}
_NAV_HINT = {
    "r2r": "Compare the History and Instruction to infer your current progress, and then select the correct "
           "direction from the candidates to go to the target location.\n",
    "reverie": "Explore the scene to find out the targeted room and object. Then select the correct direction "
               "from the candidates to go to the target location.\n",
    "soon": "Nearby areas and objects can assist you in locating the desired room and object. Select the correct direction "
            "from the candidates to go to the target location.\n",
    "cvdn": "Understand the dialog in the Instruction and infer the current progress based on the History and dialog. Then "
            "select the correct direction from the candidates to go to the target location.\n",
}
_SUM_TASK = {
    "r2r": "### Instruction: Predict the fine-grained instruction based on your previous history and current location. "
           "Fine-grained instructions contain commands for each individual step. \n",
    "reverie": "### Instruction: Generate the task you need to complete based on your previous history and current location. \n",
    "soon": "### Instruction: Generate the target you want to find based on your previous history and current location. "
            "Describe both the target and its surroundings. \n",
}
_SUM_HINT = {
    "r2r": "Please generate the step-by-step instruction.\n",
    "reverie": "Please predict the task you need to complete.\n",
    "soon": "Please predict both the target you want to find and its surroundings.\n",
}


def _hist(hist_num):
    return _HIST + "### History: {}\n".format(" ".join("({}) <hist>".format(i) for i in range(hist_num)))


def navigation_prompt(agent, instruction, hist_num, cand_num, cls_token="<cls_1>"):
    p = _NAV_TASK[agent].format(instruction) + _hist(hist_num)
    p += ("Following is the Candidate, which contains several directions you can go to at the current position, "
          "candidate (0) is stop.\n")
    p += "### Candidate: {}\n".format(" ".join("({}) <cand>".format(i) if i > 0 else "(0) stop" for i in range(cand_num)))
    p += _NAV_HINT[agent]
    p += "### Output: {}".format(cls_token)
    return p


def object_grounding_prompt(agent, instruction, hist_num, cand_num, cls_token="<cls_1>"):
    """reverie.py / soon.py get_object_grounding_prompt; cand_num = #objects + 1 (option 0 = not exist)"""
    p = "Select the target object from the candidate objects based on the instruction and history.\n"
    p += _NAV_TASK[agent].format(instruction) + _hist(hist_num)
    p += ("Following is the Object, which contains several objects that you could see at the current viewpoint, "
          "option (0) indicates not exist.\n")
    p += "### Object: {}\n".format(" ".join("({}) <cand>".format(i) if i > 0 else "(0) not exist" for i in range(cand_num)))
    p += "Select the target object from the candidate objects according to the instruction.\n"
    p += "### Output: {}".format(cls_token)
    return p


def _observation(cand_num):
    if cand_num == 0:
        return ""
    return ("Following is the Observation, which contains panoramic views at your current location.\n"
            "### Candidate: {}\n".format(" ".join("({}) <cand>".format(i) for i in range(cand_num))))


def summarization_prompt(agent, instruction, hist_num, cand_num):
    """get_summarization_prompt: the label (the instruction itself) follows '### Answer: ' (nav_model.py:299-306)"""
    return _SUM_TASK[agent] + _hist(hist_num) + _observation(cand_num) + _SUM_HINT[agent] + "### Answer: "


def embodied_qa_prompt(agent, question, hist_num, cand_num):
    """r2r.py get_embodied_qa_prompt (fine-grained R2R, mp3d_agent.py:845-868)"""
    return ("### Instruction: answer the question. \n" + (_hist(hist_num) if hist_num != 0 else "") + _observation(cand_num)
            + "### Question: {}\n".format(question) + "### Answer: ")


def qa3d_prompt(question):
    """llava.py:13-17"""
    return "### Image: <cand>\n" + "### Instruction: {}\n".format(question) + "### Output: "


def static_prefix(prompt):
    """the part of a navigation / grounding / summarization prompt that is the same at every step of an episode: everything up
    to and including "### History:" (the task sentence + the instruction + the fixed history header)"""
    key = "### History:"
    i = prompt.index(key)
    return prompt[: i + len(key)]
